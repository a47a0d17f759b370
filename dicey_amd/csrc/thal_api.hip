// dg_thal_open / dg_thal_batch: primer3's thermodynamic alignment for `dicey search`, one GPU lane per (oligo, target)
// pair.  Replaces primer3thal::get_thermodynamic_values (reference src/silica.h:320-323, src/thal.h:2374-2397) and
// primer3thal::thal() with type = thal_end1, temponly = 1 (src/silica.h:437,511; src/thal.h:2409-2655).
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#include <cmath>
#include <fstream>
#include <sstream>

#include "index_internal.hpp"
#include "thal.hpp"

#include "iupac.hpp"
#include "thal_internal.hpp"
#include "thal_wave.hpp"

namespace dg {
namespace {

// one value per line, possibly "inf" (thal.h:403-414)
struct ValueFile {
  std::ifstream f;
  bool ok;
  explicit ValueFile(const std::string& p) : f(p.c_str()), ok(f.good()) {}
  bool line(std::string& s) { return (bool)std::getline(f, s); }
  double next() {
    std::string s;
    if (!line(s)) {
      ok = false;
      return 0;
    }
    size_t k = 0;
    while (k < s.size() && std::isspace((unsigned char)s[k])) ++k;
    if (s.compare(k, 3, "inf") == 0) return thal::kInf;
    return std::strtod(s.c_str() + k, nullptr);
  }
};
static double field(const std::string& tok) { return tok == "inf" ? thal::kInf : std::strtod(tok.c_str(), nullptr); }

static int load_tables(const std::string& dir, thal::Tables& t) {
  using thal::fin;
  using thal::kInf;
  auto quad = [&](const char* sname, const char* hname, double S[5][5][5][5], double H[5][5][5][5], bool terminal) -> int {
    ValueFile fs(dir + sname), fh(dir + hname);
    if (!fs.ok || !fh.ok) return fail(DG_EIO, "cannot read %s%s / %s", dir.c_str(), sname, hname);
    for (int i = 0; i < 5; ++i)
      for (int ii = 0; ii < 5; ++ii)
        for (int j = 0; j < 5; ++j)
          for (int jj = 0; jj < 5; ++jj) {
            if (!terminal) {  // getStack / getStackint2 (thal.h:497-555)
              if (i == 4 || j == 4 || ii == 4 || jj == 4) {
                S[i][ii][j][jj] = -1.0;
                H[i][ii][j][jj] = kInf;
                continue;
              }
            } else {  // getTstack / getTstack2 (thal.h:622-679)
              if (i == 4 || j == 4) {
                H[i][ii][j][jj] = kInf;
                S[i][ii][j][jj] = -1.0;
                continue;
              }
              if (ii == 4 || jj == 4) {
                S[i][ii][j][jj] = 0.00000000001;
                H[i][ii][j][jj] = 0.0;
                continue;
              }
            }
            S[i][ii][j][jj] = fs.next();
            H[i][ii][j][jj] = fh.next();
            if (!fin(S[i][ii][j][jj]) || !fin(H[i][ii][j][jj])) {
              S[i][ii][j][jj] = -1.0;
              H[i][ii][j][jj] = kInf;
            }
          }
    if (!fs.ok || !fh.ok) return fail(DG_EFORMAT, "%s%s / %s are too short", dir.c_str(), sname, hname);
    return DG_OK;
  };
  DG_TRY(quad("stack.ds", "stack.dh", t.stackS, t.stackH, false));
  DG_TRY(quad("stackmm.ds", "stackmm.dh", t.stackmmS, t.stackmmH, false));
  {  // getDangle (thal.h:558-604): 3' block then 5' block in the same files
    ValueFile fs(dir + "dangle.ds"), fh(dir + "dangle.dh");
    if (!fs.ok || !fh.ok) return fail(DG_EIO, "cannot read %sdangle.ds/.dh", dir.c_str());
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 5; ++j)
        for (int k = 0; k < 5; ++k) {
          if (i == 4 || j == 4 || k == 4) {
            t.dangle3S[i][k][j] = -1.0;
            t.dangle3H[i][k][j] = kInf;
          } else {
            t.dangle3S[i][k][j] = fs.next();
            t.dangle3H[i][k][j] = fh.next();
            if (!fin(t.dangle3S[i][k][j]) || !fin(t.dangle3H[i][k][j])) {
              t.dangle3S[i][k][j] = -1.0;
              t.dangle3H[i][k][j] = kInf;
            }
          }
        }
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 5; ++j)
        for (int k = 0; k < 5; ++k) {
          if (i == 4 || j == 4 || k == 4) {
            t.dangle5S[i][j][k] = -1.0;
            t.dangle5H[i][j][k] = kInf;
          } else {
            t.dangle5S[i][j][k] = fs.next();
            t.dangle5H[i][j][k] = fh.next();
            if (!fin(t.dangle5S[i][j][k]) || !fin(t.dangle5H[i][j][k])) {
              t.dangle5S[i][j][k] = -1.0;
              t.dangle5H[i][j][k] = kInf;
            }
          }
        }
    if (!fs.ok || !fh.ok) return fail(DG_EFORMAT, "%sdangle.ds/.dh are too short", dir.c_str());
  }
  {  // getLoop (thal.h:606-620): "<size> <interior> <bulge> <hairpin>" per line, 30 lines
    ValueFile fs(dir + "loops.ds"), fh(dir + "loops.dh");
    if (!fs.ok || !fh.ok) return fail(DG_EIO, "cannot read %sloops.ds/.dh", dir.c_str());
    for (int k = 0; k < 30; ++k) {
      std::string ls, lh, a, b, c, d;
      if (!fs.line(ls) || !fh.line(lh)) return fail(DG_EFORMAT, "%sloops.ds/.dh are too short", dir.c_str());
      std::istringstream ss(ls), sh(lh);
      ss >> a >> b >> c >> d;
      t.interiorS[k] = field(b);
      t.bulgeS[k] = field(c);
      sh >> a >> b >> c >> d;
      t.interiorH[k] = field(b);
      t.bulgeH[k] = field(c);
    }
  }
  DG_TRY(quad("tstack_tm_inf.ds", "tstack.dh", t.tstackS, t.tstackH, true));
  DG_TRY(quad("tstack2.ds", "tstack2.dh", t.tstack2S, t.tstack2H, true));
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) {  // tableStartATS / tableStartATH (thal.h:767-783), AT_S = 6.9, AT_H = 2200
      t.atpS[i][j] = 0.00000000001;
      t.atpH[i][j] = 0.0;
    }
  t.atpS[0][3] = t.atpS[3][0] = 6.9;
  t.atpH[0][3] = t.atpH[3][0] = 2200.0;
  return DG_OK;
}

struct PairDesc {
  u64 a_off, b_off, dp_off;  // framed codes of oligo 1 / reversed oligo 2, DP planes
  u32 len1, len2;
  u32 symmetric, pad;
};

// One lane per pair, DP table in global memory: pairs too long for the wave kernel, or handed back by it (redo[t]).
__global__ void k_thal(const thal::Tables* T, thal::Env env, const PairDesc* pd, u64 n, const u8* codes, double* dp, double* temp,
                       int* end1, int* end2, const u8* redo, u32 only_redo) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const PairDesc d = pd[t];
  if (only_redo ? !redo[t] : d.pad != 0) return;  // first pass: pairs the wave kernel does not take; second: its hand-backs
  const u8* pa = codes + d.a_off;
  const u8* pb = codes + d.b_off;
  thal::Result r = thal::end1_tm<const u8*>(*T, env, pa, (int)d.len1, pb, (int)d.len2, d.symmetric != 0,
                                            reinterpret_cast<thal::Cell*>(dp + d.dp_off));
  temp[t] = r.temp;
  end1[t] = r.end1;
  end2[t] = r.end2;
}

// One wavefront per pair (thal_wave.hpp), persistent workgroups; pairs with pad == 1 only.
__global__ void __launch_bounds__(1024) k_thal_wave(const thal::Tables* T, thal::Env env, const PairDesc* pd, u64 n, const u8* codes,
                                                   double* temp, int* end1, int* end2, u8* redo, u32 len1cap, u32 stride, u32 wave_bytes,
                                                   u32 force_redo) {
  DG_DYNAMIC_LDS(lds_raw);
  thal::Tables* tabs = reinterpret_cast<thal::Tables*>(lds_raw);
  thal::wave_header_init(lds_raw, T, env);
  const thal::EndTables& ends = *thal::wave_end_tables(lds_raw);
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const u32 lane = threadIdx.x & 63;
  const thal::WaveMem wm =
      thal::wave_mem_at(lds_raw + thal::wave_header_bytes() + (size_t)wave * wave_bytes, len1cap, stride);
  for (u64 t = (u64)blockIdx.x * wpb + wave; t < n; t += (u64)gridDim.x * wpb) {
    const PairDesc d = pd[t];
    if (!d.pad) continue;
    const u32 ca = lane <= d.len1 + 1 ? codes[d.a_off + lane] : 0u;  // framed: sentinel 4 at both ends
    const u32 cb = lane <= d.len2 + 1 ? codes[d.b_off + lane] : 0u;
    thal::PlaneSeq fa, fb;
    fa.b0 = __ballot(ca & 1);
    fa.b1 = __ballot(ca & 2);
    fa.b2 = __ballot(ca & 4);
    fb.b0 = __ballot(cb & 1);
    fb.b1 = __ballot(cb & 2);
    fb.b2 = __ballot(cb & 4);
    bool amb = false;
    const thal::Result r = thal::wave_end1_tm(*tabs, ends, env, fa, (int)d.len1, fb, (int)d.len2, d.symmetric != 0, wm, (int)stride, amb);
    if (lane == 0) {
      temp[t] = r.temp;
      end1[t] = r.end1;
      end2[t] = r.end2;
      redo[t] = (amb || force_redo) ? 1 : 0;
    }
  }
}

// thal(window, reverse complement of the window) for windows of one byte buffer (`dicey padlock`: arms and probes against
// their perfect complements, padlock.h:327-371).  Oligo 2 reversed is the complement in window order, so both code
// sequences come from the same bytes: A/C/G/T -> (c, 3-c); 'U' is code 4 on the oligo and complements to 'A'
// (util.h:64); anything else is 4 on both.
struct WinDesc {
  u64 off;
  u32 len, pad;
};
// INSIDE: the end tables inside the parameter tables' LDS copy (thal_wave.hpp) — chosen by the host when it buys a wavefront per CU
template <bool INSIDE>
__global__ void __launch_bounds__(1024) k_thal_self_wave(const thal::Tables* T, thal::Env env, const WinDesc* wd, u64 n, const u8* bytes,
                                                        double* temp, u8* redo, u32 lencap, u32 wave_bytes, u32 force_redo) {
  DG_DYNAMIC_LDS(lds_raw);
  thal::Tables* tabs = reinterpret_cast<thal::Tables*>(lds_raw);
  thal::wave_header_init<INSIDE>(lds_raw, T, env);
  const thal::EndTables& ends = *thal::wave_end_tables<INSIDE>(lds_raw);
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const u32 lane = threadIdx.x & 63;
  const thal::WaveMem wm =
      thal::wave_mem_at(lds_raw + thal::wave_header_bytes<INSIDE>() + (size_t)wave * wave_bytes, lencap, lencap);
  for (u64 t = (u64)blockIdx.x * wpb + wave; t < n; t += (u64)gridDim.x * wpb) {
    const WinDesc d = wd[t];
    const u32 len = d.len;
    u32 ca = 4, cb = 4;  // frame sentinels at positions 0 and len+1
    if (lane >= 1 && lane <= len) {
      const u8 ch = bytes[d.off + lane - 1];
      ca = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
      cb = ca < 4 ? 3u - ca : (ch == 'U' ? 0u : 4u);
    }
    const bool in = lane <= len + 1;
    thal::PlaneSeq fa, fb;
    fa.b0 = __ballot(in && (ca & 1));
    fa.b1 = __ballot(in && (ca & 2));
    fa.b2 = __ballot(in && (ca & 4));
    fb.b0 = __ballot(in && (cb & 1));
    fb.b1 = __ballot(in && (cb & 2));
    fb.b2 = __ballot(in && (cb & 4));
    // symmetry_thermo (thal.h:1976-2010) of both oligos: the check is invariant under reversal, so cb stands for oligo 2
    bool sym = !(len & 1);
    if (sym) {
      bool bad = false;
      if (lane >= 1 && lane <= len / 2) {
        const u32 x = ca, y = (u32)fa[(int)(len + 1 - lane)], u = cb, v = (u32)fb[(int)(len + 1 - lane)];
        if ((x < 4 || y < 4) && (x > 3 || y > 3 || x + y != 3)) bad = true;
        if ((u < 4 || v < 4) && (u > 3 || v > 3 || u + v != 3)) bad = true;
      }
      sym = __ballot(bad) == 0;
    }
    bool amb = false;
    const thal::Result r = thal::wave_end1_tm(*tabs, ends, env, fa, (int)len, fb, (int)len, sym, wm, (int)lencap, amb);
    if (lane == 0) {
      temp[t] = r.temp;
      redo[t] = (amb || force_redo) ? 1 : 0;
    }
  }
}

static u8 code_of(char c) {
  c = (char)std::toupper((unsigned char)c);
  return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;  // str2int, thal.h:260-275
}
// symmetry_thermo (thal.h:1976-2010): even length and self-complementary
static bool self_complementary(const u8* s, size_t n) {
  if (n % 2) return false;
  for (size_t i = 0; i < n / 2; ++i) {
    char a = (char)std::toupper(s[i]), b = (char)std::toupper(s[n - 1 - i]);
    if ((a == 'A' && b != 'T') || (a == 'T' && b != 'A') || (b == 'A' && a != 'T') || (b == 'T' && a != 'A')) return false;
    if ((a == 'C' && b != 'G') || (a == 'G' && b != 'C') || (b == 'C' && a != 'G') || (b == 'G' && a != 'C')) return false;
  }
  return true;
}

}  // namespace
}  // namespace dg

using namespace dg;

extern "C" {

int dg_thal_open(const char* config_dir, double mv, double dv, double dntp, double dna_conc, int device, dg_thal** out) {
  if (!config_dir || !out) return fail(DG_EINVAL, "dg_thal_open: null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DG_ENODEV, "no HIP device available");
  if (device < 0 || device >= ndev) return fail(DG_EINVAL, "device %d out of range (have %d)", device, ndev);
  std::string dir(config_dir);
  if (!dir.empty() && dir.back() != '/') dir.push_back('/');
  dg_thal* th = new dg_thal;
  th->device = device;
  int rc = load_tables(dir, th->host_tables);
  if (rc != DG_OK) {
    delete th;
    return rc;
  }
  // saltCorrectS (thal.h:354-359) and the two RC values (thal.h:2504-2508)
  double dn = dntp;
  if (dv <= 0) dn = dv;
  th->env.salt_correction = 0.368 * ((log((mv + 120 * (sqrt(fmax(0.0, dv - dn)))) / 1000)));
  th->env.rc_sym = 1.9872 * log(dna_conc / 1000000000.0);
  th->env.rc_asym = 1.9872 * log(dna_conc / 4000000000.0);
  auto body = [&]() -> int {
    DG_HIP(hipSetDevice(device));
    DG_HIP(hipStreamCreate(&th->stream));
    DG_HIP(hipMalloc((void**)&th->d_tables, sizeof(thal::Tables)));
    DG_HIP(hipMemcpy(th->d_tables, &th->host_tables, sizeof(thal::Tables), hipMemcpyHostToDevice));
    return DG_OK;
  };
  rc = body();
  if (rc != DG_OK) {
    delete th;
    return rc;
  }
  *out = th;
  return DG_OK;
}

void dg_thal_close(dg_thal* th) {
  if (!th) return;
  (void)hipSetDevice(th->device);
  delete th;
}

int dg_thal_batch(dg_thal* th, const uint8_t* seqs, const uint64_t* off, size_t npairs, double* temp, int32_t* end1, int32_t* end2) {
  if (!th || !seqs || !off || !temp) return fail(DG_EINVAL, "dg_thal_batch: null argument");
  if (!npairs) return DG_OK;
  static const bool timing = std::getenv("DICEY_TIMING") != nullptr;
  auto t_start = std::chrono::steady_clock::now();
  std::vector<PairDesc> pd(npairs);
  u64 ncode = 0, ndp = 0;
  static const bool no_wave = exp_env("DICEY_NO_WAVE_THAL") != nullptr;       // debugging aids: sequential kernel only /
  static const bool force_redo = exp_env("DICEY_DEBUG_THAL_REDO") != nullptr;  // recompute every pair sequentially too
  const u64 kWaveLenCap = 48;  // two 48 x 48 tables still fit a workgroup's LDS next to the parameter tables
  u32 wl1 = 0, wl2 = 0;
  u64 nwave = 0;
  static u8 lut[256];
  static bool lut_ready = false;
  if (!lut_ready) {
    for (int c = 0; c < 256; ++c) lut[c] = code_of((char)c);
    lut_ready = true;
  }
  for (size_t k = 0; k < npairs; ++k) {
    u64 l1 = off[2 * k + 1] - off[2 * k], l2 = off[2 * k + 2] - off[2 * k + 1];
    if (l1 > 10000 || l2 > 10000) return fail(DG_ELIMIT, "pair %zu: sequence longer than THAL_MAX_SEQ", k);
    pd[k].len1 = (u32)l1;
    pd[k].len2 = (u32)l2;
    pd[k].a_off = ncode;
    ncode += l1 + 2;
    pd[k].b_off = ncode;
    ncode += l2 + 2;
    pd[k].symmetric = self_complementary(seqs + off[2 * k], l1) && self_complementary(seqs + off[2 * k + 1], l2);
    pd[k].pad = (!no_wave && l1 >= 1 && l2 >= 1 && l1 <= kWaveLenCap && l2 <= kWaveLenCap) ? 1 : 0;  // wave kernel takes it
    pd[k].dp_off = ndp;  // the sequential kernel's table; hand-backs of the wave kernel get theirs in the second pass
    const bool both_long = l1 > (u64)thal::kMaxAlign && l2 > (u64)thal::kMaxAlign;
    if (!pd[k].pad) ndp += both_long ? 0 : 2 * l1 * l2;
    if (pd[k].pad) {
      wl1 = std::max<u32>(wl1, (u32)l1);
      wl2 = std::max<u32>(wl2, (u32)l2);
      ++nwave;
    }
  }
  std::vector<u8> codes(ncode);
  for (size_t k = 0; k < npairs; ++k) {
    const u8* s1 = seqs + off[2 * k];
    const u8* s2 = seqs + off[2 * k + 1];
    u8* ca = codes.data() + pd[k].a_off;
    u8* cb = codes.data() + pd[k].b_off;
    const u32 l1 = pd[k].len1, l2 = pd[k].len2;
    ca[0] = ca[l1 + 1] = cb[0] = cb[l2 + 1] = 4;
    for (u32 i = 0; i < l1; ++i) ca[1 + i] = lut[s1[i]];
    for (u32 j = 0; j < l2; ++j) cb[1 + j] = lut[s2[l2 - 1 - j]];  // reversed
  }
  if (timing) std::fprintf(stderr, "dg_thal_batch: host prep %.1f ms for %zu pairs\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), npairs);
  DG_HIP(hipSetDevice(th->device));
  hipStream_t st = th->stream;
  DG_TRY(th->ws[0].reserve(npairs * sizeof(PairDesc)));
  DG_TRY(th->ws[1].reserve(ncode + 8));
  DG_TRY(th->ws[2].reserve((ndp + 1) * 8));
  DG_TRY(th->ws[3].reserve(npairs * 8));
  DG_TRY(th->ws[4].reserve(npairs * 8));
  DG_HIP(hipMemcpyAsync(th->ws[0].p, pd.data(), npairs * sizeof(PairDesc), hipMemcpyHostToDevice, st));
  DG_HIP(hipMemcpyAsync(th->ws[1].p, codes.data(), ncode, hipMemcpyHostToDevice, st));
  int* e1 = th->ws[4].as<int>();
  int* e2 = e1 + npairs;
  DG_TRY(th->ws[5].reserve(npairs + 8));
  DG_HIP(hipMemsetAsync(th->ws[5].p, 0, npairs, st));
  if (nwave) {
    const u32 tab_bytes = thal::wave_header_bytes();
    const u32 per_wave = thal::wave_mem_bytes(wl1, wl2);
    const u32 lds_cap = 160 * 1024;
    const u32 wpb = std::max<u32>(1, std::min<u32>(16, (lds_cap - tab_bytes) / per_wave));
    const u32 lds_total = thal::wave_lds_total(wpb, per_wave);
    int cus = 0;
    DG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, th->device));
    DG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_thal_wave), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total));
    const u64 blocks = std::min<u64>(ceil_div(npairs, wpb), (u64)cus * std::max<u32>(1, lds_cap / lds_total));
    hipLaunchKernelGGL(k_thal_wave, dim3((u32)blocks), dim3(wpb * 64), lds_total, st, (const thal::Tables*)th->d_tables, th->env,
                       th->ws[0].as<PairDesc>(), (u64)npairs, th->ws[1].as<u8>(), th->ws[3].as<double>(), e1, e2, th->ws[5].as<u8>(), wl1,
                       wl2, per_wave, force_redo ? 1u : 0u);
  }
  if (nwave < npairs)
    hipLaunchKernelGGL(k_thal, dim3(ceil_div(npairs, 64)), dim3(64), 0, st, (const thal::Tables*)th->d_tables, th->env,
                       th->ws[0].as<PairDesc>(), (u64)npairs, th->ws[1].as<u8>(), th->ws[2].as<double>(), th->ws[3].as<double>(), e1, e2,
                       (const u8*)th->ws[5].as<u8>(), 0u);
  std::vector<u8> redo(npairs, 0);
  if (nwave) {  // pairs the wave kernel handed back go through the sequential kernel, with tables allocated only now
    DG_HIP(hipMemcpyAsync(redo.data(), th->ws[5].p, npairs, hipMemcpyDeviceToHost, st));
    DG_HIP(hipStreamSynchronize(st));
    u64 ndp2 = 0, nredo = 0;
    for (size_t k = 0; k < npairs; ++k)
      if (redo[k]) {
        pd[k].dp_off = ndp2;
        ndp2 += 2 * (u64)pd[k].len1 * pd[k].len2;
        ++nredo;
      }
    if (nredo) {
      DG_TRY(th->ws[2].reserve((ndp2 + 1) * 8));
      DG_HIP(hipMemcpyAsync(th->ws[0].p, pd.data(), npairs * sizeof(PairDesc), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_thal, dim3(ceil_div(npairs, 64)), dim3(64), 0, st, (const thal::Tables*)th->d_tables, th->env,
                         th->ws[0].as<PairDesc>(), (u64)npairs, th->ws[1].as<u8>(), th->ws[2].as<double>(), th->ws[3].as<double>(), e1, e2,
                         (const u8*)th->ws[5].as<u8>(), 1u);
    }
  }
  DG_HIP(hipMemcpyAsync(temp, th->ws[3].p, npairs * 8, hipMemcpyDeviceToHost, st));
  std::vector<int> h1(npairs), h2(npairs);
  DG_HIP(hipMemcpyAsync(h1.data(), e1, npairs * 4, hipMemcpyDeviceToHost, st));
  DG_HIP(hipMemcpyAsync(h2.data(), e2, npairs * 4, hipMemcpyDeviceToHost, st));
  DG_HIP(hipStreamSynchronize(st));
  DG_HIP(hipGetLastError());
  for (size_t k = 0; k < npairs; ++k) {
    if (end1) end1[k] = h1[k];
    if (end2) end2[k] = h2[k];
  }
  return DG_OK;
}

}  // extern "C"

// thal(window, its reverse complement) for n windows of `bytes` (host), every window at most kSelfWindowMax nt: the pairs
// are formed on the device (k_thal_self_wave).  Hand-backs of the wave kernel are recomputed through dg_thal_batch.
// win_len == nullptr: every window has `uniform_len` characters (r06: `padlock` passed a vector of 1.5 M equal lengths)
int dg::thal_self_windows(dg_thal* th, const uint8_t* bytes, uint64_t nbytes, const uint64_t* win_off, const uint32_t* win_len, size_t n,
                          double* temp, uint32_t uniform_len) {
  if (!n) return DG_OK;
  static const bool force_redo = exp_env("DICEY_DEBUG_THAL_REDO") != nullptr;
  // the descriptors are written by up to eight host threads into a buffer this thread keeps between calls (r06: 24 MB per 1.5 M
  // windows were value-initialised and filled by one thread, ~8 ms of a 0.27 s padlock step)
  static thread_local std::vector<WinDesc> wd;
  if (wd.size() < n) wd.resize(n);
  std::atomic<u32> maxlen_a{0};
  std::atomic<size_t> bad_at{(size_t)-1};
  {
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(8, n / 65536));
    WinDesc* const wdp = wd.data();
    auto fill = [&, wdp](size_t k0, size_t k1) {
      u32 mx = 0;
      for (size_t k = k0; k < k1; ++k) {
        const u32 len = win_len ? win_len[k] : uniform_len;
        if (len == 0 || len > kSelfWindowMax || win_off[k] + len > nbytes) {
          size_t cur = bad_at.load();
          while (k < cur && !bad_at.compare_exchange_weak(cur, k)) {
          }
          return;
        }
        wdp[k].off = win_off[k];
        wdp[k].len = len;
        wdp[k].pad = 0;
        mx = std::max(mx, len);
      }
      u32 cur = maxlen_a.load();
      while (mx > cur && !maxlen_a.compare_exchange_weak(cur, mx)) {
      }
    };
    if (nt <= 1) fill(0, n);
    else {
      std::vector<std::thread> pool;
      for (unsigned t = 0; t < nt; ++t) pool.emplace_back(fill, n * t / nt, n * (t + 1) / nt);
      for (auto& x : pool) x.join();
    }
  }
  if (bad_at.load() != (size_t)-1) return fail(DG_EINVAL, "thal_self_windows: window %zu out of range", bad_at.load());
  const u32 maxlen = maxlen_a.load();
  static const bool no_wave = exp_env("DICEY_NO_WAVE_THAL") != nullptr;  // debugging aid: sequential kernel only
  static thread_local std::vector<u8> redo;
  redo.assign(n, 1);
  if (!no_wave) {
    DG_HIP(hipSetDevice(th->device));
    hipStream_t st = th->stream;
    DG_TRY(th->ws[0].reserve(n * sizeof(WinDesc)));
    DG_TRY(th->ws[1].reserve(nbytes + 8));
    DG_TRY(th->ws[3].reserve(n * 8));
    DG_TRY(th->ws[5].reserve(n + 8));
    DG_HIP(hipMemcpyAsync(th->ws[0].p, wd.data(), n * sizeof(WinDesc), hipMemcpyHostToDevice, st));
    DG_HIP(hipMemcpyAsync(th->ws[1].p, bytes, nbytes, hipMemcpyHostToDevice, st));
    const u32 per_wave = thal::wave_mem_bytes(maxlen, maxlen);
    const u32 lds_cap = 160 * 1024;
    auto waves_with = [&](u32 header) { return std::max<u32>(1, std::min<u32>(16, (lds_cap - header) / per_wave)); };
    // the end tables inside the parameter tables when that buys a wavefront per CU (40-mers: 7 instead of 6), else where they always were
    const bool inside = waves_with(thal::wave_header_bytes<true>()) > waves_with(thal::wave_header_bytes<false>());
    const u32 wpb = waves_with(inside ? thal::wave_header_bytes<true>() : thal::wave_header_bytes<false>());
    const u32 lds_total = inside ? thal::wave_lds_total<true>(wpb, per_wave) : thal::wave_lds_total<false>(wpb, per_wave);
    int cus = 0;
    DG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, th->device));
    const u64 blocks = std::min<u64>(ceil_div(n, wpb), (u64)cus * std::max<u32>(1, lds_cap / lds_total));
    if (inside) {
      DG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_thal_self_wave<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total));
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_thal_self_wave<true>), dim3((u32)blocks), dim3(wpb * 64), lds_total, st, (const thal::Tables*)th->d_tables, th->env,
                         th->ws[0].as<WinDesc>(), (u64)n, th->ws[1].as<u8>(), th->ws[3].as<double>(), th->ws[5].as<u8>(), maxlen, per_wave,
                         force_redo ? 1u : 0u);
    } else {
      DG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_thal_self_wave<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total));
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_thal_self_wave<false>), dim3((u32)blocks), dim3(wpb * 64), lds_total, st, (const thal::Tables*)th->d_tables, th->env,
                         th->ws[0].as<WinDesc>(), (u64)n, th->ws[1].as<u8>(), th->ws[3].as<double>(), th->ws[5].as<u8>(), maxlen, per_wave,
                         force_redo ? 1u : 0u);
    }
    DG_HIP(hipMemcpyAsync(temp, th->ws[3].p, n * 8, hipMemcpyDeviceToHost, st));
    DG_HIP(hipMemcpyAsync(redo.data(), th->ws[5].p, n, hipMemcpyDeviceToHost, st));
    DG_HIP(hipStreamSynchronize(st));
    DG_HIP(hipGetLastError());
  }
  std::vector<size_t> again;
  for (size_t k = 0; k < n; ++k)
    if (redo[k]) again.push_back(k);
  if (!again.empty()) {  // explicit pairs through the general entry point (whose wave kernel hands them to the sequential one)
    std::string buf;
    std::vector<uint64_t> off(1, 0);
    for (size_t k : again) {
      const uint8_t* w = bytes + win_off[k];
      const uint32_t len = win_len ? win_len[k] : uniform_len;
      buf.append((const char*)w, len);
      off.push_back(buf.size());
      for (uint32_t i = 0; i < len; ++i) buf.push_back(complement_iupac((char)w[len - 1 - i]));
      off.push_back(buf.size());
    }
    std::vector<double> t2(again.size());
    int rc = dg_thal_batch(th, (const uint8_t*)buf.data(), off.data(), again.size(), t2.data(), nullptr, nullptr);
    if (rc != DG_OK) return rc;
    for (size_t i = 0; i < again.size(); ++i) temp[again[i]] = t2[i];
  }
  return DG_OK;
}
