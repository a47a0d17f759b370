// dg_index_open / dg_index_close / dg_index_stats: load the sdsl csa_wt<> file unchanged into HBM and derive the
// search layouts on the device.  Replaces load_from_checked_file (reference src/hunter.h:253-260, src/silica.h:340-347).
#include <chrono>

#include "devfm.hpp"
#include "index_internal.hpp"
#include <rocprim/device/device_scan.hpp>

#include "sdsl_file.hpp"

namespace dg {

std::string& last_error() {
  static thread_local std::string e;
  return e;
}
int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

// ------------------------------------------------------------------------------------------------------------
// Derivation kernels (load time).  All are one-thread-one-chain; no inter-thread communication.
// ------------------------------------------------------------------------------------------------------------

// Thread h decodes BWT[64h, 64h+64) through the wavelet tree into three plane words and per-symbol counts.
__global__ void k_decode_bwt(FmView f, OccBlock* occ, u32* half_cnt, u64 nhalf) {
  u64 h = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= nhalf) return;
  u64 p0 = 0, p1 = 0, p2 = 0;
  u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  u64 base = h << 6;
  for (u32 j = 0; j < 64; ++j) {
    u64 i = base + j;
    if (i >= f.n) break;
    u32 sym;
    (void)wt_inverse_select(f, i, sym);
    u32 code = code_of_byte(sym);
    p0 |= (u64)(code & 1) << j;
    p1 |= (u64)((code >> 1) & 1) << j;
    p2 |= (u64)(code >> 2) << j;
    c0 += code == 0;
    c1 += code == 1;
    c2 += code == 2;
    c3 += code == 3;
  }
  OccBlock* b = occ + (h >> 1);
  u32 w = (u32)(h & 1);
  b->pl[0][w] = p0;
  b->pl[1][w] = p1;
  b->pl[2][w] = p2;
  half_cnt[h] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);  // each <= 64
}

// Thread t sums HALF_PER_CHUNK half-block counts -> chunk totals (host scans the chunk totals).
static constexpr u32 HALF_PER_CHUNK = 64;
__global__ void k_chunk_totals(const u32* half_cnt, u64 nhalf, u32* chunk_tot /*[nchunk][4]*/, u64 nchunk) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nchunk) return;
  u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  u64 b = t * HALF_PER_CHUNK, e = b + HALF_PER_CHUNK < nhalf ? b + HALF_PER_CHUNK : nhalf;
  for (u64 h = b; h < e; ++h) {
    u32 v = half_cnt[h];
    s0 += v & 255;
    s1 += (v >> 8) & 255;
    s2 += (v >> 16) & 255;
    s3 += v >> 24;
  }
  chunk_tot[t * 4 + 0] = s0;
  chunk_tot[t * 4 + 1] = s1;
  chunk_tot[t * 4 + 2] = s2;
  chunk_tot[t * 4 + 3] = s3;
}
// chunk_base = exclusive scan of chunk_tot; writes the running counts into every block header
__global__ void k_block_counts(const u32* half_cnt, u64 nhalf, const u32* chunk_base, u64 nchunk, OccBlock* occ,
                               u64 nblk) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nchunk) return;
  u32 s0 = chunk_base[t * 4], s1 = chunk_base[t * 4 + 1], s2 = chunk_base[t * 4 + 2], s3 = chunk_base[t * 4 + 3];
  u64 b = t * HALF_PER_CHUNK, e = b + HALF_PER_CHUNK;
  for (u64 h = b; h < e; ++h) {
    if (!(h & 1) && (h >> 1) < nblk) {
      OccBlock* ob = occ + (h >> 1);
      ob->cnt[0] = s0;
      ob->cnt[1] = s1;
      ob->cnt[2] = s2;
      ob->cnt[3] = s3;
    }
    if (h < nhalf) {
      u32 v = half_cnt[h];
      s0 += v & 255;
      s1 += (v >> 8) & 255;
      s2 += (v >> 16) & 255;
      s3 += v >> 24;
    }
  }
}

// Text from ISA samples: sample k is ISA[64k]; walking LF from it yields T[64k-1], T[64k-2], ...
// Thread 0 starts at ISA[0] and therefore produces the tail T[n-1] .. T[64*(ns-1)].
__global__ void k_derive_text(FmView f, u8* text) {
  u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= f.n_isa_samp) return;
  u64 i = packed_get(f.isa_samp, f.samp_width, k);
  u64 hi, lo;  // writes text[lo, hi)
  if (k == 0) {
    hi = f.n;
    lo = (f.n_isa_samp - 1) * 64;
  } else {
    hi = k * 64;
    lo = hi - 64;
  }
  for (u64 p = hi; p > lo;) {
    u32 sym;
    i = lf_step(f, i, sym);
    text[--p] = (u8)sym;
  }
}

// The same walk also knows the rank of every suffix it passes: after the LF step that produced T[p], i is the rank of
// suffix p.  isa[p] = i is written next to text[p] (consecutive p per lane), and the suffix array is obtained from it by
// a sort (derive_sa_by_sort) instead of a second walk with scattered stores.
__global__ void k_derive_text_isa(FmView f, u8* text, u32* isa) {
  u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= f.n_isa_samp) return;
  u64 i = packed_get(f.isa_samp, f.samp_width, k);
  u64 hi, lo;  // writes [lo, hi)
  if (k == 0) {
    hi = f.n;
    lo = (f.n_isa_samp - 1) * 64;
  } else {
    hi = k * 64;
    lo = hi - 64;
  }
  for (u64 p = hi; p > lo;) {
    u32 sym;
    i = lf_step(f, i, sym);
    --p;
    text[p] = (u8)sym;
    isa[p] = (u32)i;
  }
}
// the file's own SA samples must agree with the suffix array derived from its ISA samples
__global__ void k_check_sa_samples(FmView f, const u32* sa, u32* bad) {
  u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= f.n_sa_samp) return;
  if (packed_get(f.sa_samp, f.samp_width, j) != sa[j * 32]) atomicAdd(bad, 1u);
}

// Full SA from SA samples: SA[LF(i)] = SA[i] - 1.  Thread j owns the LF chain that leaves index 32j and stops at
// the next index that is itself sampled; LF is a permutation, so every unsampled index is written exactly once.
__global__ void k_derive_sa(FmView f, u32* sa) {
  u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= f.n_sa_samp) return;
  u64 v = packed_get(f.sa_samp, f.samp_width, j);
  u64 i = j * 32;
  sa[i] = (u32)v;
  for (;;) {
    u32 sym;
    i = lf_step(f, i, sym);
    if ((i & 31) == 0) break;
    v = v ? v - 1 : f.n - 1;
    sa[i] = (u32)v;
  }
}

// Self-check on a pseudo-random sample of SA indices: BWT[i] == T[SA[i]-1], and suffix SA[i] <= suffix SA[i+1]
// on their first 48 bytes.  Catches a mis-parsed file before any query is answered.
__global__ void k_selfcheck(FmView f, u64 nsample, u32* bad) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nsample) return;
  u64 x = (t + 1) * 0x9E3779B97F4A7C15ULL;
  x ^= x >> 29;
  x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 32;
  u64 i = x % f.n;
  u32 sym;
  (void)lf_step(f, i, sym);
  u64 p = f.sa[i];
  u64 prev = p ? p - 1 : f.n - 1;
  bool ok = f.text[prev] == (u8)sym && p < f.n;
  if (i + 1 < f.n) {
    u64 q = f.sa[i + 1];
    for (u32 k = 0; k < 48; ++k) {
      u64 a = p + k, b = q + k;
      if (a >= f.n || b >= f.n) break;
      u8 ca = f.text[a], cb = f.text[b];
      if (ca != cb) {
        ok = ok && ca < cb;
        break;
      }
    }
  }
  if (!ok) atomicAdd(bad, 1u);
}

// K-mer table: thread i looks at the K-mer that starts suffix SA[i]; the first / last suffix of each run of equal K-mers
// writes the interval bounds.  K-mers that run into a separator, an N or the end of the text have no entry.
// the K-mer and the K2-mer (K2 > K, or 0) that start at text position p (K, K2 <= 24).  The text arrives as four aligned
// 64-bit words (24 characters at any byte offset; the buffer is padded) in one round trip — a byte loop that may stop at the
// first other letter cannot have its loads hoisted, so it waited for memory once per character (r02: 0.45-0.70 s of table
// fill).  Which bytes are A/C/G/T is an exact zero-byte test against the four letters; ((b >> 1) ^ (b >> 2)) & 3 maps
// A,C,G,T to 0..3; the 2-bit fields of a byte-swapped word are squeezed together so that the first character ends up on top.
DG_DEV u64 acgt_flags(u64 w) {  // 0x80 in every byte that is one of A, C, G, T
  u64 f = 0;
  const u64 lo7 = 0x7F7F7F7F7F7F7F7FULL;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u64 v = w ^ ((u64)(k == 0 ? 'A' : k == 1 ? 'C' : k == 2 ? 'G' : 'T') * 0x0101010101010101ULL);
    f |= ~(((v & lo7) + lo7) | v | lo7);
  }
  return f;
}
DG_DEV u32 pack8(u64 w) {  // 8 characters (first one in the lowest byte) -> 16 bits, first character in bits 15-14
  w = __builtin_bswap64(w);
  u64 x = ((w >> 1) ^ (w >> 2)) & 0x0303030303030303ULL;
  x = (x | (x >> 6)) & 0x000F000F000F000FULL;
  x = (x | (x >> 12)) & 0x000000FF000000FFULL;
  x = (x | (x >> 24)) & 0xFFFFULL;
  return (u32)x;
}
DG_DEV void kmer_codes_at(const FmView& f, u64 p, u32 K, u32 K2, u64& code, u64& code2) {
  const u64 none = 1ULL << 63;
  const u64* src = reinterpret_cast<const u64*>(f.text + (p & ~7ULL));
  const u32 sh = (u32)(p & 7) * 8;
  const u64 w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
  const u64 b0 = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
  const u64 b1 = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1;
  const u64 b2 = sh ? (w2 >> sh) | (w3 << (64 - sh)) : w2;
  // bit i of `valid` <=> character i is A/C/G/T
  const u64 gather = 0x0102040810204080ULL, ones = 0x0101010101010101ULL;
  const u32 v0 = (u32)((((acgt_flags(b0) >> 7) & ones) * gather) >> 56), v1 = (u32)((((acgt_flags(b1) >> 7) & ones) * gather) >> 56),
            v2 = (u32)((((acgt_flags(b2) >> 7) & ones) * gather) >> 56);
  const u32 valid = v0 | (v1 << 8) | (v2 << 16);
  u32 got = (u32)__builtin_ctz(~valid | (1u << 24));  // characters before the first other letter, at most 24
  const u64 room = f.n - 1 - p;                        // characters before the sentinel (p < n - 1 for every suffix but the last)
  if (p >= f.n - 1) got = 0;
  else if (got > room) got = (u32)room;
  const u64 acc = (u64)pack8(b0) << 32 | (u64)pack8(b1) << 16 | (u64)pack8(b2);  // 24 characters, the first on top (bits 47-46)
  code = got >= K ? acc >> (2 * (24 - K)) : none;
  code2 = (K2 && got >= K2) ? acc >> (2 * (24 - K2)) : none;
}
// kf2: copy 0 of the long presence filter (bit address = K2-mer code), zeroed by the caller; nullptr = none
__global__ void k_kmer_table(FmView f, uint2* tab, u32 K, u32 K2, u32* kf2) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= f.n) return;
  u64 me, me2, prev, next, x2;
  kmer_codes_at(f, f.sa[i], K, kf2 ? K2 : 0u, me, me2);
  if (me >> 63) return;
  prev = next = 1ULL << 63;
  if (i) kmer_codes_at(f, f.sa[i - 1], K, 0u, prev, x2);
  if (i + 1 < f.n) kmer_codes_at(f, f.sa[i + 1], K, 0u, next, x2);
  if (prev != me) tab[me].x = (u32)i;
  if (next != me) tab[me].y = (u32)(i + 1);
  if (kf2 && !(me2 >> 63)) atomicOr(&kf2[me2 >> 5], 1u << (me2 & 31));
}

// ---- presence filter (FmView::kf) ----
// copy 0 (in-line bits = the low 9 code bits = address order of the table): one coalesced pass over the table
__global__ void __launch_bounds__(256) k_kf_from_table(const uint2* tab, u64 entries, u32* out) {  // entries: multiple of 64
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < entries; i += stride) {  // a launch holds < 2^32 threads
    const uint2 e = tab[i];
    const unsigned long long m = __ballot(e.x < e.y);
    if ((threadIdx.x & 63) == 0) *reinterpret_cast<unsigned long long*>(out + (i >> 5)) = m;
  }
}
// any copy, one lane per output word, straight from the table: small tables, and the lane-independent debugging build
__global__ void k_kf_generic(const uint2* tab, u32 bits, u32 s, u32* out, u64 nwords) {
  const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  const u64 line = w >> 4;
  u32 v = 0;
  for (u32 i = 0; i < 32; ++i) {
    const u64 inl = ((w & 15) << 5) | i;
    const u64 code = (line & ((1ULL << s) - 1)) | (inl << s) | ((line >> s) << (s + 9));
    if (code >> bits) continue;
    const uint2 e = tab[code];
    v |= (u32)(e.x < e.y) << i;
  }
  out[w] = v;
}
// copy with in-line bits [s, s+9), s >= 8, from copy 0: a workgroup owns the 2^17 codes that share everything but those
// nine bits and the low eight; lane M reads the 256 bits of its in-line value (32 contiguous bytes of copy 0), the
// 512 x 256 bit matrix is turned by ballots and leaves as 256 consecutive lines of the new copy.
__global__ void __launch_bounds__(512) k_kf_transpose(const u32* r0, u32* out, u32 s) {
  __shared__ unsigned long long tile[256][8];  // [line][wavefront]: 64 in-line bits each
  const u64 T = blockIdx.x;
  const u64 lowpart = (T & ((1ULL << (s - 8)) - 1)) << 8, high = T >> (s - 8);
  const u32 M = threadIdx.x, wave = M >> 6, lane = M & 63;
  const u64 src_bit = lowpart | ((u64)M << s) | (high << (s + 9));
  const uint4* src = reinterpret_cast<const uint4*>(r0 + (src_bit >> 5));
  const uint4 a = src[0], b = src[1];
  const u32 word[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned long long keep[4] = {0, 0, 0, 0};
#pragma unroll
  for (int wi = 0; wi < 8; ++wi)
#pragma unroll
    for (int bit = 0; bit < 32; ++bit) {
      const unsigned long long m = __ballot((word[wi] >> bit) & 1u);
      const int l = wi * 32 + bit;
      if ((int)lane == (l & 63)) keep[l >> 6] = m;
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) tile[64 * k + lane][wave] = keep[k];
  __syncthreads();
  const u64 line0 = lowpart | (high << s);
  unsigned long long* dst = reinterpret_cast<unsigned long long*>(out + line0 * 16);
  const unsigned long long* flat = &tile[0][0];
  for (u32 k = threadIdx.x; k < 256 * 8; k += 512) dst[k] = flat[k];
}

// any copy, one lane per output word, from copy 0 (in-line bits = the low nine code bits, i.e. bit address = code): the long
// filter has no table to read
__global__ void k_kf_generic0(const u32* r0, u32 bits, u32 s, u32* out, u64 nwords) {
  const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  const u64 line = w >> 4;
  u32 v = 0;
  for (u32 i = 0; i < 32; ++i) {
    const u64 inl = ((w & 15) << 5) | i;
    const u64 code = (line & ((1ULL << s) - 1)) | (inl << s) | ((line >> s) << (s + 9));
    if (code >> bits) continue;
    v |= ((r0[code >> 5] >> (code & 31)) & 1u) << i;
  }
  out[w] = v;
}

__global__ void k_kf_compare(const u32* a, const u32* b, u64 n, u32* bad) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && a[i] != b[i]) atomicAdd(bad, 1u);
}

// Number of copies, their in-line bit fields (spread evenly over the code) and, per window position, the copy to ask.
static void kf_layout(KFilter& kf, u32 k) {
  const u32 bits = 2 * k;
  u32 nr = bits <= 9 ? 1u : (bits - 9 + 7) / 8 + 1;
  if (nr > 4) nr = 4;
  kf.k = k;
  kf.nr = nr;
  for (u32 r = 0; r < 4; ++r) kf.s[r] = (nr > 1 && r < nr) ? (r * (bits - 9) + (nr - 1) / 2) / (nr - 1) : 0u;
  kf.pick = 0;
  for (u32 t = 0; t < 32; ++t) {
    u32 best = 0;
    int best_score = -1000;
    for (u32 r = 0; r < nr; ++r) {
      const int lo = (int)kf.s[r], hi = lo + 9, b0 = 2 * (int)t;
      const int covered = (b0 >= lo && b0 < hi) + (b0 + 1 >= lo && b0 + 1 < hi);
      const int off = 2 * b0 + 1 - (2 * lo + 8);  // twice the distance of the character from the field's centre
      const int score = covered * 100 - (off < 0 ? -off : off);
      if (score > best_score) {
        best_score = score;
        best = r;
      }
    }
    kf.pick |= (u64)best << (2 * t);
  }
}

// Copies of a presence filter.  Copy 0 comes from the table (tab != nullptr: one ballot pass) or is already in place
// (copy0: the long filter, whose bits the table-fill kernel sets); the others are bit-matrix transposes of copy 0.
static int build_filter(dg_index* ix, KFilter& kf, u32 k, const uint2* tab, u32* copy0) {
  const u32 bits = 2 * k;
  if (bits < 17 || exp_env("DICEY_NO_KMER_FILTER")) return DG_OK;  // a table this small is cache resident anyway
  const u64 entries = 1ULL << bits, nwords = entries >> 5;
  kf_layout(kf, k);
  const u32 nr = kf.nr;
  kf.nr = 0;  // until the copies exist
  const u32 TB = 256;
  for (u32 r = 0; r < nr; ++r) {
    const u32 s = kf.s[r];
    u32* bm = nullptr;
    if (r == 0 && copy0) bm = copy0;
    else {
      DG_HIP(big_alloc((void**)&bm, nwords * 4 + 64, ix->stream));
      ix->owned.push_back(bm);
      ix->hbm_bytes += nwords * 4 + 64;
    }
    if (r == 0 && copy0) {
      // filled by the caller
    } else if (r == 0) {
      hipLaunchKernelGGL(k_kf_from_table, dim3((u32)std::min<u64>(entries / TB, 1u << 20)), dim3(TB), 0, ix->stream, tab, entries, bm);
    } else if (s < 8) {
      if (tab) hipLaunchKernelGGL(k_kf_generic, dim3(ceil_div(nwords, TB)), dim3(TB), 0, ix->stream, tab, bits, s, bm, nwords);
      else hipLaunchKernelGGL(k_kf_generic0, dim3(ceil_div(nwords, TB)), dim3(TB), 0, ix->stream, (const u32*)kf.cp[0], bits, s, bm, nwords);
    } else {
      hipLaunchKernelGGL(k_kf_transpose, dim3((u32)(entries >> 17)), dim3(512), 0, ix->stream, kf.cp[0], bm, s);
    }
    kf.cp[r] = bm;
  }
  DG_HIP(hipStreamSynchronize(ix->stream));
  DG_HIP(hipGetLastError());
  kf.nr = nr;
  return DG_OK;
}

// The same table with one text gather per lane instead of three: the K-mers of the neighbouring suffixes come from the
// neighbouring lanes (the first / last lane of a wavefront read theirs).
__global__ void __launch_bounds__(256) k_kmer_table_wave(FmView f, uint2* tab, u32 K, u32 K2, u32* kf2) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 none = 1ULL << 63;
  u64 me = none, me2 = none, x2;
  if (i < f.n) kmer_codes_at(f, f.sa[i], K, kf2 ? K2 : 0u, me, me2);
  const u32 lane = threadIdx.x & 63;
  u32 plo = __shfl_up((u32)me, 1), phi = __shfl_up((u32)(me >> 32), 1);
  u32 nlo = __shfl_down((u32)me, 1), nhi = __shfl_down((u32)(me >> 32), 1);
  u64 prev = (u64)phi << 32 | plo, next = (u64)nhi << 32 | nlo;
  const u64 prev2 = (u64)__shfl_up((u32)(me2 >> 32), 1) << 32 | __shfl_up((u32)me2, 1);
  if (i >= f.n || (me >> 63)) return;
  if (lane == 0) {
    prev = none;
    if (i) kmer_codes_at(f, f.sa[i - 1], K, 0u, prev, x2);
  }
  if (lane == 63) {
    next = none;
    if (i + 1 < f.n) kmer_codes_at(f, f.sa[i + 1], K, 0u, next, x2);
  }
  if (prev != me) tab[me].x = (u32)i;
  if (next != me) tab[me].y = (u32)(i + 1);
  // suffixes are sorted, so equal K2-mers are neighbours: the first lane of a run sets the bit
  if (kf2 && !(me2 >> 63) && (lane == 0 || prev2 != me2)) atomicOr(&kf2[me2 >> 5], 1u << (me2 & 31));
}

// FmView::nrun_min: the length of the text's shortest run of 'N', capped at 64 (r05).  A lane per text position; the lane of a run's
// first character looks at most 63 characters ahead.  A query character outside A,C,G,T is searched as 'N' (util.h:208-219) and
// matches only an 'N' of the text: a neighbourhood string whose N-block has other characters on both sides needs a run of exactly
// that length — k_search prunes such strings when no run is that short (hunt_search.hpp).
__global__ void __launch_bounds__(256) k_nrun_min(const u8* text, u64 n, u32* out) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    if (text[i] != 'N' || (i && text[i - 1] == 'N')) continue;
    u32 len = 1;
    while (len < 64 && i + len < n && text[i + len] == 'N') ++len;
    if (len < 64) atomicMin(out, len);
  }
}

// FmView::pre5: the five characters in front of every suffix, in suffix-array order (one text gather per lane) — and, with the same
// gather, FmView::sax: {SA[i], context word} (devfm.hpp: the two characters in front of the suffix and the thirteen characters
// T[p+16 .. p+28], two bits each, bit 31 when any of them is not A/C/G/T or lies outside the text).  sax == nullptr: pre5 only.
__global__ void __launch_bounds__(256) k_pre5(const u32* sa, const u8* text, u64 n, u16* out, uint2* sax) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 p = sa[i];
  u32 v = 0;
#pragma unroll
  for (u32 k = 1; k <= 5; ++k) {
    u32 c = 7u;
    if (p >= k) {
      const u32 b = text[p - k];
      c = b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 7u;
    }
    v |= c << (3 * (k - 1));
  }
  out[i] = (u16)v;
  if (!sax) return;
  const u32 c1 = v & 7u, c2 = (v >> 3) & 7u;
  u32 ctx = (c1 & 3u) | ((c2 & 3u) << 2);
  bool esc = c1 > 3u || c2 > 3u;
  const u64 q = p + SAX_POST_OFF;
  if (q + SAX_POST_N > n) esc = true;  // (text[n - 1] is the sentinel: it escapes by itself)
  else {
    // thirteen bytes from q, read as three aligned 64-bit words (the text block is 64-byte aligned and 64 bytes longer than n)
    const u64* tw = reinterpret_cast<const u64*>(text) + (q >> 3);
    const u32 sh = (u32)(q & 7u) * 8u;
    const u64 a = tw[0], b = tw[1], c = tw[2];
    const u64 w0 = sh ? (a >> sh) | (b << (64u - sh)) : a, w1 = sh ? (b >> sh) | (c << (64u - sh)) : b;
#pragma unroll
    for (u32 j = 0; j < SAX_POST_N; ++j) {
      const u32 ch = (u32)((j < 8 ? w0 >> (8 * j) : w1 >> (8 * (j - 8))) & 255u);
      const u32 code = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 7u;
      esc = esc || code > 3u;
      ctx |= (code & 3u) << (4 + 2 * j);
    }
  }
  sax[i] = make_uint2((u32)p, esc ? (ctx | SAX_ESCAPE) : ctx);
}

// FmView::plv, one level: the flags "text position below x" of every suffix-array index as 64-bit words (seven per 64-byte
// line of the rank directory, one wavefront per line), the line's count aside ...
__global__ void __launch_bounds__(256) k_plv_bits(const u32* sa, u64 n, u64 x, u64* dir, u32* cnt, u64 nlines) {
  const u64 line = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  if (line >= nlines) return;  // (the same for a whole wavefront)
  u32 total = 0;
  u64 mine = 0;
  for (u32 j = 0; j < 7; ++j) {
    const u64 i = line * FmView::PLV_LINE + 64 * j + lane;
    const unsigned long long w = __ballot(i < n && (u64)sa[i] < x);
    if (lane == j) mine = w;
    total += (u32)__popcll(w);
  }
  if (lane < 7) dir[line * 8 + 1 + lane] = mine;
  if (lane == 0) cnt[line] = total;
}
// ... and, behind the exclusive scan of the counts: the count before every line, and the flagged suffixes' records in their order
__global__ void __launch_bounds__(256) k_plv_fill(const uint2* sax, const u32* before, u64* dir, uint2* rec, u64 nlines) {
  const u64 line = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  if (line >= nlines) return;
  u64 base = before[line];
  if (lane == 0) dir[line * 8] = base;
  for (u32 j = 0; j < 7; ++j) {
    const u64 w = dir[line * 8 + 1 + j];
    if ((w >> lane) & 1ULL) rec[base + (u64)__popcll(w & ((1ULL << lane) - 1ULL))] = sax[line * FmView::PLV_LINE + 64 * j + lane];
    base += (u64)__popcll(w);
  }
}

// The table's final entry format (FmView::ktab): (lo, hi) as the fill pass left them -> (lo, width | pre5[lo] << 16) for widths below
// 2^16, (lo, 0x80000000 | width) above.  pre: nullptr when the preceding characters were not built (the field then reads 0x7FFF,
// which the search kernels never consult without FmView::pre5).
__global__ void __launch_bounds__(256) k_ktab_pack(uint2* tab, u64 entries, const u16* pre) {
  // (grid-stride: a launch of entries / 256 workgroups is 2^32 threads for the order-16 table, one more than HIP takes)
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < entries; i += (u64)gridDim.x * blockDim.x) {
    const uint2 e = tab[i];
    const u32 w = e.y - e.x;  // (absent K-mers are (0, 0))
    u32 y;
    if (w >= 65536u) y = 0x80000000u | w;
    else y = w | ((w && pre ? (u32)pre[e.x] & 0x7FFFu : 0x7FFFu) << 16);
    tab[i].y = y;
  }
}

// Block minima of the suffix array, fan-out 8 (FmView::samin): one lane per block, two 16-byte loads.
__global__ void __launch_bounds__(256) k_block_min8(const u32* in, u64 n_in, u32* out, u64 n_out, u64 n_out_padded) {
  const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_out_padded) return;
  u32 m = 0xFFFFFFFFu;
  if (b < n_out) {
    const u64 i0 = b * 8;
    if (i0 + 8 <= n_in) {
      const uint4 x = *reinterpret_cast<const uint4*>(in + i0), y = *reinterpret_cast<const uint4*>(in + i0 + 4);
      m = min(min(min(x.x, x.y), min(x.z, x.w)), min(min(y.x, y.y), min(y.z, y.w)));
    } else {
      for (u64 i = i0; i < n_in; ++i) m = min(m, in[i]);
    }
  }
  out[b] = m;
}

struct PhaseClock {  // DICEY_TIMING=1: host wall clock of the load phases on stderr
  bool on = std::getenv("DICEY_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char* what) {
    if (!on) return;
    auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "dicey timing: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

static int derive_layouts(dg_index* ix, const SdslCsa& c, u32 flags) {
  FmView& f = ix->view;
  PhaseClock pc;
  const u64 n = f.n;
  const u64 nblk = (n >> 7) + 1, nhalf = nblk * 2;
  const u64 nchunk = (nhalf + HALF_PER_CHUNK - 1) / HALF_PER_CHUNK;
  OccBlock* occ = nullptr;
  u32 *half_cnt = nullptr, *chunk = nullptr, *sa = nullptr, *bad = nullptr;
  u8* text = nullptr;
  DG_HIP(big_alloc((void**)&occ, nblk * sizeof(OccBlock), ix->stream));
  ix->owned.push_back(occ);
  DG_HIP(hipMemsetAsync(occ, 0, nblk * sizeof(OccBlock), ix->stream));
  DG_HIP(big_alloc((void**)&half_cnt, nhalf * 4, ix->stream));
  DG_HIP(hipMalloc((void**)&chunk, nchunk * 16));
  const u32 TB = 256;
  hipLaunchKernelGGL(k_decode_bwt, dim3(ceil_div(nhalf, TB)), dim3(TB), 0, ix->stream, f, occ, half_cnt, nhalf);
  hipLaunchKernelGGL(k_chunk_totals, dim3(ceil_div(nchunk, TB)), dim3(TB), 0, ix->stream, half_cnt, nhalf, chunk, nchunk);
  std::vector<u32> tot(nchunk * 4);
  DG_HIP(hipMemcpyAsync(tot.data(), chunk, nchunk * 16, hipMemcpyDeviceToHost, ix->stream));
  DG_HIP(hipStreamSynchronize(ix->stream));
  u64 run[4] = {0, 0, 0, 0};
  for (u64 t = 0; t < nchunk; ++t)
    for (int s = 0; s < 4; ++s) {
      u32 v = tot[t * 4 + s];
      tot[t * 4 + s] = (u32)run[s];
      run[s] += v;
    }
  // C[] of the file must agree with what the decoded BWT holds
  static const u8 acgt[4] = {'A', 'C', 'G', 'T'};
  for (int s = 0; s < 4; ++s) {
    u32 cc = c.char2comp[acgt[s]];
    u64 expect = cc ? c.C[cc + 1] - c.C[cc] : 0;  // comp 0 is the sentinel: the byte is absent
    if (run[s] != expect)
      return fail(DG_EFORMAT, "index self-check: decoded BWT holds %llu '%c' but C[] says %llu", (unsigned long long)run[s],
                  acgt[s], (unsigned long long)expect);
    f.C4[s] = cc ? (u32)c.C[cc] : 0u;
  }
  DG_HIP(hipMemcpyAsync(chunk, tot.data(), nchunk * 16, hipMemcpyHostToDevice, ix->stream));
  hipLaunchKernelGGL(k_block_counts, dim3(ceil_div(nchunk, TB)), dim3(TB), 0, ix->stream, half_cnt, nhalf, chunk, nchunk, occ,
                     nblk);
  f.occ = occ;
  DG_HIP(hipStreamSynchronize(ix->stream));
  big_free(half_cnt, ix->stream);
  DG_HIP(hipFree(chunk));
  pc.lap("occ blocks");

  DG_HIP(big_alloc((void**)&text, n + 64, ix->stream));
  ix->owned.push_back(text);
  DG_HIP(big_alloc((void**)&sa, n * 4 + 64, ix->stream));
  ix->owned.push_back(sa);
  bool sorted_sa = false;
  if (n > (1u << 16)) {
    // text and inverse suffix array in one walk, suffix array by sorting (isa[p], p)
    u32* isa = nullptr;
    if (big_alloc((void**)&isa, n * 4 + 64, ix->stream) == hipSuccess) {
      hipLaunchKernelGGL(k_derive_text_isa, dim3(ceil_div(f.n_isa_samp, TB)), dim3(TB), 0, ix->stream, f, text, isa);
      const int rc = derive_sa_by_sort(ix->stream, isa, n, sa);
      big_free(isa, ix->stream);
      if (rc == DG_OK) {
        u32* badp = nullptr;
        u32 hbad = 0;
        DG_HIP(hipMalloc((void**)&badp, 4));
        DG_HIP(hipMemsetAsync(badp, 0, 4, ix->stream));
        hipLaunchKernelGGL(k_check_sa_samples, dim3(ceil_div(f.n_sa_samp, TB)), dim3(TB), 0, ix->stream, f, (const u32*)sa, badp);
        DG_HIP(hipMemcpyAsync(&hbad, badp, 4, hipMemcpyDeviceToHost, ix->stream));
        DG_HIP(hipStreamSynchronize(ix->stream));
        DG_HIP(hipFree(badp));
        if (hbad) return fail(DG_EFORMAT, "index self-check: %u suffix-array samples of the file disagree with its inverse samples", hbad);
        sorted_sa = true;
      } else if (rc != DG_ENODEV && rc != DG_ENOMEM) return rc;
    } else (void)hipGetLastError();
  }
  if (!sorted_sa) {  // the two independent walks (small indexes, low memory, debugging builds)
    hipLaunchKernelGGL(k_derive_text, dim3(ceil_div(f.n_isa_samp, TB)), dim3(TB), 0, ix->stream, f, text);
    hipLaunchKernelGGL(k_derive_sa, dim3(ceil_div(f.n_sa_samp, TB)), dim3(TB), 0, ix->stream, f, sa);
  }
  f.text = text;
  f.sa = sa;
  DG_HIP(hipStreamSynchronize(ix->stream));
  DG_HIP(hipGetLastError());
  pc.lap("text + suffix array");
  {  // the shortest run of 'N' (one streaming pass over the text)
    u32* d_min = nullptr;
    u32 h_min = 64;
    DG_HIP(hipMalloc((void**)&d_min, 4));
    DG_HIP(hipMemcpyAsync(d_min, &h_min, 4, hipMemcpyHostToDevice, ix->stream));
    hipLaunchKernelGGL(k_nrun_min, dim3((u32)std::min<u64>(ceil_div(n, TB), 1u << 16)), dim3(TB), 0, ix->stream, (const u8*)text, n, d_min);
    DG_HIP(hipMemcpyAsync(&h_min, d_min, 4, hipMemcpyDeviceToHost, ix->stream));
    DG_HIP(hipStreamSynchronize(ix->stream));
    DG_HIP(hipFree(d_min));
    f.nrun_min = exp_env("DICEY_NO_NRUN_PRUNE") ? 0u : h_min;
    pc.lap("shortest N run");
  }
  if (!(flags & DG_OPEN_NO_KMER_TABLE)) {
    // K = ceil(log4 n) clamped to [8,16]: about one expected occurrence per K-mer; 16 -> 34 GB, which is what the
    // 288 GB of HBM are for
    u32 K = 8;
    while (K < 16 && (1ULL << (2 * K)) < n) ++K;
    // What the rest of the HBM is spent on (DESIGN.md "long filter"): one character more for the table when the device has
    // room (K = 17, 137 GB, on a 3.1 Gb genome: one interval extension fewer per surviving string), and a presence filter of
    // order K2 = ceil(log4 n) + 2 in four permuted copies (4 x 8.6 GB at K2 = 18) in front of everything — a random 18-mer of
    // a 3.1 Gb genome occurs with p = 0.04, a 16-mer with 0.51, a 17-mer with 0.17.  r02 measured K / K2 = 16/19, 16/18 and
    // 17/18 on the bench workloads; 17/18 is the fastest at both distances.
    u32 K2 = 0;
    {
      size_t free_b = 0, total_b = 0;
      const bool have = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
      const u64 slack = (u64)std::min<u64>(48ULL << 30, n * 16 + (64u << 20));
      const u64 f2_b = 4 * ((1ULL << (2 * (K + 2))) >> 3);
      u64 need = (8ULL << (2 * K)) + slack;
      if (have && free_b > need + f2_b) {
        K2 = K + 2;
        need += f2_b;
      }
      if ((flags & DG_OPEN_BIG_TABLE) && have && (1ULL << (2 * K)) < n * 2 && free_b > need - (8ULL << (2 * K)) + (8ULL << (2 * (K + 1)))) ++K;
    }
    if (const char* ek = std::getenv("DICEY_KMER_K")) {  // tuning knobs: force the table order (8..17) / the long filter's (0 = none)
      int v = std::atoi(ek);
      if (v >= 8 && v <= 17) {
        K = (u32)v;
        if (K2 && K2 <= K) K2 = K + 1;
      }
    }
    if (const char* ek = std::getenv("DICEY_KMER_K2")) {
      int v = std::atoi(ek);
      K2 = (v > (int)K && v <= 20) ? (u32)v : 0u;
    }
    if (exp_env("DICEY_NO_KMER_FILTER") || 2 * K2 < 17) K2 = 0;
    pc.lap("table order (hipMemGetInfo)");
    uint2* tab = nullptr;
    u64 entries = 1ULL << (2 * K);
    DG_HIP(big_alloc((void**)&tab, entries * sizeof(uint2), ix->stream));
    ix->owned.push_back(tab);
    pc.lap("table hipMalloc");
    DG_HIP(hipMemsetAsync(tab, 0, entries * sizeof(uint2), ix->stream));
    u32* f2 = nullptr;  // copy 0 of the long filter: its bits are set by the pass that fills the table
    if (K2) {
      const u64 f2_bytes = ((1ULL << (2 * K2)) >> 3) + 64;
      DG_HIP(big_alloc((void**)&f2, f2_bytes, ix->stream));
      ix->owned.push_back(f2);
      ix->hbm_bytes += f2_bytes;
      DG_HIP(hipMemsetAsync(f2, 0, f2_bytes, ix->stream));
    }
    hipLaunchKernelGGL(k_kmer_table_wave, dim3(ceil_div(n, TB)), dim3(TB), 0, ix->stream, f, tab, K, K2, f2);
    DG_HIP(hipStreamSynchronize(ix->stream));
    DG_HIP(hipGetLastError());
    f.ktab = tab;
    f.K = K;
    ix->hbm_bytes += entries * sizeof(uint2);
    pc.lap("table fill");
    DG_TRY(build_filter(ix, f.kf, K, tab, nullptr));
    pc.lap("presence filter");
    if (K2) {
      DG_TRY(build_filter(ix, f.kf2, K2, nullptr, f2));
      pc.lap("long presence filter");
    }
    // the characters in front of every suffix (2 n bytes): built with the table, i.e. for handles that search batches
    if (!(flags & DG_OPEN_NO_PRE5) && !exp_env("DICEY_NO_PRE5")) {
      u16* pre = nullptr;
      if (big_alloc((void**)&pre, n * 2 + 64, ix->stream) == hipSuccess) {
        ix->owned.push_back(pre);
        ix->hbm_bytes += n * 2 + 64;
        // with the same gather: the suffix array with context (8 n bytes, FmView::sax) for the locate job kernels — handles that
        // have the block minima (i.e. the top-k locate) and the room; DICEY_NO_SAX leaves it out (the tests run both ways: hits of
        // repeat-rich strings then read their context from the text as before r06)
        uint2* sax = nullptr;
        if (!(flags & DG_OPEN_COMPACT) && !exp_env("DICEY_NO_SAX") && !exp_env("DICEY_NO_SA_MINIMA")) {
          size_t free_b = 0, total_b = 0;
          if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > n * 8 + n * 2 + (4ULL << 30) &&
              big_alloc((void**)&sax, n * 8 + 64, ix->stream) == hipSuccess) {
            ix->owned.push_back(sax);
            ix->hbm_bytes += n * 8 + 64;
          } else {
            sax = nullptr;
            (void)hipGetLastError();
          }
        }
        hipLaunchKernelGGL(k_pre5, dim3(ceil_div(n, TB)), dim3(TB), 0, ix->stream, (const u32*)sa, (const u8*)text, n, pre, sax);
        DG_HIP(hipStreamSynchronize(ix->stream));
        DG_HIP(hipGetLastError());
        f.pre5 = pre;
        f.sax = sax;
        pc.lap(sax ? "preceding characters + suffix array with context" : "preceding characters");
        // prefix levels over the records (FmView::plv): X = 2^16, 2^18, ... below n — 64 n / 448 bytes of directory and 8 X bytes
        // of records per level (GRCh38: 8 levels, 3.5 + 11.5 GB); DICEY_NO_PLV leaves them out (tests: every repeat-rich string
        // then walks the block minima)
        if (sax && !exp_env("DICEY_NO_PLV")) {
          const u64 nlines = n / FmView::PLV_LINE + 1;
          u32 *cnt = nullptr, *before = nullptr;
          void* tmp = nullptr;
          size_t tmp_bytes = 0;
          (void)rocprim::exclusive_scan(nullptr, tmp_bytes, (const u32*)nullptr, (u32*)nullptr, 0u, nlines, rocprim::plus<u32>(), ix->stream);
          if (hipMalloc((void**)&cnt, nlines * 4) == hipSuccess && hipMalloc((void**)&before, nlines * 4) == hipSuccess &&
              hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16) == hipSuccess) {
            for (u32 lv = 0; lv < FmView::MAXPLV; ++lv) {
              const u64 x = 1ULL << (16 + 2 * lv);
              if (x >= n) break;
              size_t free_b = 0, total_b = 0;
              if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < nlines * 64 + x * 8 + (4ULL << 30)) break;
              u64* dir = nullptr;
              uint2* rec = nullptr;
              if (big_alloc((void**)&dir, nlines * 64, ix->stream) != hipSuccess) break;
              ix->owned.push_back(dir);
              if (big_alloc((void**)&rec, x * 8 + 64, ix->stream) != hipSuccess) break;
              ix->owned.push_back(rec);
              ix->hbm_bytes += nlines * 64 + x * 8 + 64;
              hipLaunchKernelGGL(k_plv_bits, dim3(ceil_div(nlines, 4)), dim3(256), 0, ix->stream, (const u32*)sa, n, x, dir, cnt, nlines);
              if (rocprim::exclusive_scan(tmp, tmp_bytes, (const u32*)cnt, before, 0u, nlines, rocprim::plus<u32>(), ix->stream) != hipSuccess) break;
              hipLaunchKernelGGL(k_plv_fill, dim3(ceil_div(nlines, 4)), dim3(256), 0, ix->stream, (const uint2*)sax, (const u32*)before, dir, rec, nlines);
              DG_HIP(hipStreamSynchronize(ix->stream));
              DG_HIP(hipGetLastError());
              f.plv[lv].dir = dir;
              f.plv[lv].rec = rec;
              f.plv[lv].x = x;
              f.nplv = lv + 1;
            }
          }
          (void)hipGetLastError();
          if (cnt) (void)hipFree(cnt);
          if (before) (void)hipFree(before);
          if (tmp) (void)hipFree(tmp);
          pc.lap("prefix levels of the suffix array");
        }
      } else (void)hipGetLastError();
    }
    // every reader of (lo, hi) pairs is done (the filters above were derived from them): the entries take their final form
    hipLaunchKernelGGL(k_ktab_pack, dim3((u32)std::min<u64>(ceil_div(entries, TB), 1u << 20)), dim3(TB), 0, ix->stream, tab, entries, f.pre5);
    DG_HIP(hipStreamSynchronize(ix->stream));
    DG_HIP(hipGetLastError());
    pc.lap("table entries packed");
  }
  // block minima over the suffix array for the top-k locate (0.57 n bytes; one streaming pass over SA)
  f.samin[0] = sa;
  f.nlev = 1;
  if (!exp_env("DICEY_NO_SA_MINIMA")) {
    u64 cnt = n;
    const u32* prev = sa;
    while (cnt > 8 && f.nlev < FmView::MAXLEV) {
      const u64 nout = (cnt + 7) / 8, npad = ((nout + 7) & ~7ULL) + 8;
      u32* lv = nullptr;
      DG_HIP(big_alloc((void**)&lv, npad * 4, ix->stream));
      ix->owned.push_back(lv);
      ix->hbm_bytes += npad * 4;
      hipLaunchKernelGGL(k_block_min8, dim3(ceil_div(npad, TB)), dim3(TB), 0, ix->stream, prev, cnt, lv, nout, npad);
      f.samin[f.nlev++] = lv;
      prev = lv;
      cnt = nout;
    }
    DG_HIP(hipStreamSynchronize(ix->stream));
    DG_HIP(hipGetLastError());
    pc.lap("suffix-array block minima");
  }
  if (!(flags & DG_OPEN_NO_SELFCHECK)) {
    DG_HIP(hipMalloc((void**)&bad, 4));
    DG_HIP(hipMemsetAsync(bad, 0, 4, ix->stream));
    u64 ns = n < (1u << 20) ? n : (1u << 20);
    hipLaunchKernelGGL(k_selfcheck, dim3(ceil_div(ns, TB)), dim3(TB), 0, ix->stream, f, ns, bad);
    u32 hbad = 0;
    DG_HIP(hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, ix->stream));
    DG_HIP(hipStreamSynchronize(ix->stream));
    DG_HIP(hipFree(bad));
    if (hbad) return fail(DG_EFORMAT, "index self-check failed on %u of %llu sampled suffixes", hbad, (unsigned long long)ns);
    pc.lap("self-check");
  }
  ix->hbm_bytes += nblk * sizeof(OccBlock) + n + 64 + n * 4 + 64;
  return DG_OK;
}

template <class T>
static int upload(dg_index* ix, const void* src, size_t bytes, const T** dst) {
  void* d = nullptr;
  DG_HIP(big_alloc(&d, bytes + 64, ix->stream));  // +64: packed_get may touch one word past the last element
  ix->owned.push_back(d);
  DG_HIP(hipMemsetAsync((char*)d + bytes, 0, 64, ix->stream));
  DG_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ix->stream));
  ix->hbm_bytes += bytes + 64;
  *dst = (const T*)d;
  return DG_OK;
}

static int open_impl(const char* path, int device, u32 flags, dg_index* ix) {
  auto t0 = std::chrono::steady_clock::now();
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DG_ENODEV, "no HIP device available");
  if (device < 0 || device >= ndev) return fail(DG_EINVAL, "device %d out of range (have %d)", device, ndev);
  DG_HIP(hipSetDevice(device));
  ix->device = device;
  DG_HIP(hipStreamCreate(&ix->stream));
  PhaseClock pc;
  SdslCsa c;
  DG_TRY(sdsl_open(path, c));
  pc.lap("file map + parse");
  if (c.n > 0xFFFFFFFFULL) return fail(DG_ELIMIT, "index has %llu symbols; this build keeps 32-bit suffix-array entries", (unsigned long long)c.n);
  ix->file_bytes = c.len;
  FmView& f = ix->view;
  std::memset(&f, 0, sizeof f);
  f.n = c.n;
  DG_TRY(upload(ix, c.bv.w, c.bv.nwords * 8, &f.bv));
  DG_TRY(upload(ix, c.rank.w, c.rank.nwords * 8, &f.rk));
  DG_TRY(upload(ix, c.sa_samples.w, c.sa_samples.nwords * 8, &f.sa_samp));
  DG_TRY(upload(ix, c.isa_samples.w, c.isa_samples.nwords * 8, &f.isa_samp));
  f.samp_width = c.sa_samples.width;
  f.n_sa_samp = c.sa_samples.bits / c.sa_samples.width;
  f.n_isa_samp = c.isa_samples.bits / c.isa_samples.width;
  WtTables* t = new WtTables;
  std::memset(t, 0, sizeof *t);
  for (size_t v = 0; v < c.nodes.size(); ++v) {
    t->node_pos[v] = c.nodes[v].bv_pos;
    t->node_rank[v] = c.nodes[v].bv_pos_rank;
    t->child[v][0] = c.nodes[v].child[0];
    t->child[v][1] = c.nodes[v].child[1];
  }
  std::memcpy(t->path, c.path, sizeof t->path);
  std::memcpy(t->c_to_leaf, c.c_to_leaf, sizeof t->c_to_leaf);
  for (size_t i = 0; i < c.C.size(); ++i) t->C[i] = c.C[i];
  std::memcpy(t->char2comp, c.char2comp, 256);
  for (size_t i = 0; i < c.comp2char.size(); ++i) t->comp2char[i] = c.comp2char[i];
  t->sigma = c.sigma;
  int rc = upload(ix, t, sizeof *t, &f.wt);
  for (int b = 0; b < 256; ++b) ix->code_len[b] = (c.c_to_leaf[b] == 0xFFFF) ? 0u : (u32)(c.path[b] >> 56);
  ix->sigma = c.sigma;
  if (rc != DG_OK) {
    delete t;
    return rc;
  }
  DG_HIP(hipStreamSynchronize(ix->stream));
  delete t;
  pc.lap("upload of the sdsl sections");
  static const u8 soc[8] = {'A', 'C', 'G', 'T', 'N', '\n', 0, 0};
  std::memcpy(f.sym_of_code, soc, 8);
  auto t1 = std::chrono::steady_clock::now();
  DG_TRY(derive_layouts(ix, c, flags));
  auto t2 = std::chrono::steady_clock::now();
  ix->derive_seconds = std::chrono::duration<double>(t2 - t1).count();
  ix->load_seconds = std::chrono::duration<double>(t2 - t0).count();
  return DG_OK;
}

}  // namespace dg

using namespace dg;

dg_index::~dg_index() {
  stop_worker();
  if (lanes[0]) {  // the owner of the internal lanes (they only point at its hints record)
    for (dg_index*& l : lanes) {
      delete l;
      l = nullptr;
    }
    SharedHints* sh = shared_hints.exchange(nullptr);
    if (sh)
      for (hipEvent_t e : sh->ev_base)
        if (e) (void)hipEventDestroy(e);
    delete sh;
  }
  for (void* p : owned) dg::big_free(p, stream);
  if (stream) (void)hipStreamSynchronize(stream);
  for (auto& w : ws) w.release();
  for (auto& e : ev)
    if (e) (void)hipEventDestroy(e);

  if (pinned) (void)hipHostFree(pinned);
  if (stream) (void)hipStreamDestroy(stream);
}

extern "C" {

const char* dg_last_error(void) { return dg::last_error().c_str(); }
int dg_abi_version(void) { return DG_ABI_VERSION; }

// Host only (no device is touched): the file's sections accounted for byte by byte, their invariants against each other, and —
// DG_FM9_CHECK_DEEP — the big sections read through (rank words against the bit vector, the tree's node sizes and rank offsets,
// C[] against the leaf sizes, sample values).  The report is one JSON object; on DG_EFORMAT its "error" (= dg_last_error()) names
// the first section that is off.
int dg_fm9_check(const char* fm9_path, uint32_t flags, char* report, size_t report_cap) {
  if (report && report_cap) report[0] = 0;
  if (!fm9_path) return fail(DG_EINVAL, "dg_fm9_check: null path");
  SdslCsa c;
  int rc = sdsl_open(fm9_path, c);
  if (rc == DG_OK && (flags & DG_FM9_CHECK_DEEP)) {
    rc = sdsl_check_deep(c);
    if (rc != DG_OK) fail(rc, "%s: %s", fm9_path, c.why.c_str());
  }
  if (report && report_cap) {
    std::string j = "{\"ok\":";
    j += rc == DG_OK ? "true" : "false";
    j += ",\"deep\":";
    j += (flags & DG_FM9_CHECK_DEEP) ? "true" : "false";
    j += ",\"file_bytes\":" + std::to_string(c.len);
    if (c.base && rc != DG_EIO) {
      j += std::string(",\"layout\":\"") + (c.checked_layout ? "store_to_checked_file" : "store_to_file") + "\"";
      j += ",\"n\":" + std::to_string(c.n) + ",\"sigma\":" + std::to_string(c.wt_sigma);
      j += ",\"sa_sample_width\":" + std::to_string((unsigned)c.sa_samples.width);
      j += ",\"sections\":[";
      for (size_t i = 0; i < c.sections.size(); ++i) {
        if (i) j += ",";
        j += std::string("{\"name\":\"") + c.sections[i].name + "\",\"offset\":" + std::to_string(c.sections[i].offset) + ",\"bytes\":" + std::to_string(c.sections[i].bytes) + "}";
      }
      j += "]";
    }
    if (rc != DG_OK) {
      std::string e = last_error();
      for (char& ch : e)
        if (ch == '"' || ch == '\\' || (unsigned char)ch < 32) ch = ' ';
      j += ",\"error\":\"" + e + "\"";
    }
    j += "}";
    std::snprintf(report, report_cap, "%s", j.c_str());
  }
  return rc;
}
int dg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int dg_index_open(const char* fm9_path, int device, uint32_t flags, dg_index** out) {
  if (!fm9_path || !out) return fail(DG_EINVAL, "dg_index_open: null argument");
  *out = nullptr;
  dg_index* ix = new dg_index;
  int rc = open_impl(fm9_path, device, flags, ix);
  if (rc != DG_OK) {
    delete ix;
    return rc;
  }
  *out = ix;
  return DG_OK;
}

int dg_index_share(dg_index* src, dg_index** out) {
  if (!src || !out) return fail(DG_EINVAL, "dg_index_share: null argument");
  *out = nullptr;
  DG_HIP(hipSetDevice(src->device));
  dg_index* ix = new dg_index;
  ix->device = src->device;
  ix->view = src->view;  // the index arrays stay owned by `src`
  ix->sigma = src->sigma;
  std::memcpy(ix->code_len, src->code_len, sizeof ix->code_len);
  ix->file_bytes = src->file_bytes;
  ix->hbm_bytes = 0;
  ix->shard_cap_hint = src->shard_cap_hint;
  ix->hit_cap_hint = src->hit_cap_hint;
  ix->flat_cap_hint = src->flat_cap_hint;
  ix->generic_hint = src->generic_hint;
  ix->jobs_hint = src->jobs_hint;
  ix->fetch_hits_hint = src->fetch_hits_hint;
  ix->fused_leaves_hint = src->fused_leaves_hint;
  // DICEY_EXP_PRIO (measurement aid, r06): the shares of an index — its internal lanes — get stream priorities in turn (highest,
  // lowest, highest, ...) while the index's own stream stays at the default, so that the lanes' search kernels do not share the chip
  // evenly (which keeps the lanes in step: DESIGN "lanes")
  static std::atomic<unsigned> n_shares{0};
  hipError_t se = hipSuccess;
  if (exp_env("DICEY_EXP_PRIO")) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    se = hipStreamCreateWithPriority(&ix->stream, hipStreamDefault, (n_shares.fetch_add(1) & 1u) ? least : greatest);
  } else se = hipStreamCreate(&ix->stream);
  if (se != hipSuccess) {
    delete ix;
    return fail(DG_EHIP, "dg_index_share: cannot create a stream");
  }
  *out = ix;
  return DG_OK;
}

void* dg_index_stream(dg_index* ix) { return ix ? (void*)ix->stream : nullptr; }

void dg_index_close(dg_index* ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  delete ix;
}

int dg_index_stats(const dg_index* ix, dg_index_stats_t* out) {
  if (!ix || !out) return fail(DG_EINVAL, "dg_index_stats: null argument");
  std::memset(out, 0, sizeof *out);
  out->n = ix->view.n;
  out->sigma = ix->sigma;
  std::memcpy(out->code_len, ix->code_len, sizeof out->code_len);
  out->file_bytes = ix->file_bytes;
  out->hbm_bytes = ix->hbm_bytes;
  out->load_seconds = ix->load_seconds;
  out->derive_seconds = ix->derive_seconds;
  return DG_OK;
}

}  // extern "C"
