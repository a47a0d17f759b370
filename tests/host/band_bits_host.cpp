// Test harness (not part of the product): the verify stage's bit-plane alignment (dicey_amd/csrc/band_bits.hpp) compiled for the
// host, so that `pytest -m "not gpu"` can hold it against the checker's needle() on random windows.
#include <vector>

#include "../../dicey_amd/csrc/band_bits.hpp"

// key_on: the window comes from (key, pre, post) — the hit's own codes, band_window_from_key — instead of the text (r06)
extern "C" int bb_align_kw(const uint8_t* text, uint64_t text_n, uint64_t loc, uint32_t mlen, uint32_t n, uint32_t d, int indel, int wide,
                           const uint32_t* masks /* a c g t */, uint32_t* info, uint32_t* ops /* [2] */, uint32_t* pre_eff, int key_on, uint64_t key,
                           uint32_t pre, uint32_t post) {
  std::vector<uint32_t> tr(64, 0);
  uint64_t win[8] = {0};
  uint32_t fault = 0;
  dg::PosMasks pm{masks[0], masks[1], masks[2], masks[3]};
  dg::AlnRes r;
  const dg::KeyWindow kwv{key, pre, post};
  if (wide) r = dg::band_align_bits<13, uint32_t, 1>(text, text_n, indel != 0, loc, mlen, n, d, pm, tr.data(), win, fault, key_on != 0, kwv);
  else {
    std::vector<uint16_t> t16(64, 0);
    r = dg::band_align_bits<7, uint16_t, 1>(text, text_n, indel != 0, loc, mlen, n, d, pm, t16.data(), win, fault, key_on != 0, kwv);
  }
  *info = r.info;
  ops[0] = r.op[0];
  ops[1] = r.op[1];
  *pre_eff = r.pre_eff;
  return (int)fault;
}
extern "C" int bb_align(const uint8_t* text, uint64_t text_n, uint64_t loc, uint32_t mlen, uint32_t n, uint32_t d, int indel, int wide,
                        const uint32_t* masks /* a c g t */, uint32_t* info, uint32_t* ops /* [2] */, uint32_t* pre_eff) {
  return bb_align_kw(text, text_n, loc, mlen, n, d, indel, wide, masks, info, ops, pre_eff, 0, 0, 0, 0);
}
