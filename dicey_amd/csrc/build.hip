// dg_index_build / dg_index_build_device: construct the FM-index of a genome text on the GPU and store it in the
// sdsl csa_wt<> file layout — what `dicey index` does with sdsl::construct + store_to_checked_file
// (reference src/index.h:97-123).  The text is SEQ1 '\n' SEQ2 '\n' ... SEQk '\n', upper-case, no NUL.
//
// Pipeline (all arrays in HBM; sized for 288 GB so a 3.1 Gb genome is one pass, no partitioning):
//   1. byte histogram -> order-preserving dense codes (sentinel 0 is code 0)
//   2. suffix array by prefix doubling: keys = first K symbols packed into 64 bits, device radix sort (rocPRIM),
//      then rounds of (group rank, rank of suffix i+h) sorts restricted to still-ambiguous groups
//   3. BWT gather; Huffman-shaped wavelet tree: per inner node one wavefront-ballot pass that appends the
//      member symbols' path bits; rank_support_v blocks; SA/ISA samples
//   4. host: select_support_mcl structures, byte_tree, serialisation
// rocPRIM (header-only, ships with ROCm) is used for the radix sort and scans of this tool; the search path
// (hunt.hip) uses no library.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <chrono>

#include "index_internal.hpp"
#include "sdsl_writer.hpp"

namespace dg {
namespace {

struct Scratch {  // one resizable temp buffer for rocPRIM
  void* p = nullptr;
  size_t cap = 0;
  int need(size_t b) {
    if (b <= cap) return DG_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    DG_HIP(hipMalloc(&p, b));
    cap = b;
    return DG_OK;
  }
  ~Scratch() {
    if (p) (void)hipFree(p);
  }
};

template <class T>
struct DArr {
  T* p = nullptr;
  u64 n = 0;
  int alloc(u64 count) {
    release();
    n = count;
    DG_HIP(hipMalloc((void**)&p, (count ? count : 1) * sizeof(T)));
    return DG_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DArr() { release(); }
};

__global__ void k_histogram(const u8* text, u64 len, unsigned long long* hist) {
  __shared__ u32 h[256];
  for (u32 i = threadIdx.x; i < 256; i += blockDim.x) h[i] = 0;
  __syncthreads();
  u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) atomicAdd(&h[text[i]], 1u);
  __syncthreads();
  for (u32 i = threadIdx.x; i < 256; i += blockDim.x)
    if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

struct CodeMap {
  u8 code[256];
};

// key[i] = codes of S[i .. i+K) packed most-significant first; S = text + sentinel(code 0); zero padded past the end
__global__ void k_init_keys(const u8* text, u64 n, CodeMap cm, u32 bps, u32 K, u64* key, u32* idx) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 k = 0;
  for (u32 j = 0; j < K; ++j) {
    u64 p = i + j;
    u64 c = p < n - 1 ? cm.code[text[p]] : 0;
    k = (k << bps) | c;
  }
  key[i] = k;
  idx[i] = (u32)i;
}

// head[j] = j if element j starts a new run of equal keys, else 0  (max-scan turns it into "start of my run")
__global__ void k_run_heads(const u64* key, u64 n, u32* head) {
  u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  head[j] = (j == 0 || key[j] != key[j - 1]) ? (u32)j : 0u;
}
__global__ void k_group_heads(const u64* key, u64 n, u32* head) {  // runs of equal HIGH word (the old group)
  u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  head[j] = (j == 0 || (key[j] >> 32) != (key[j - 1] >> 32)) ? (u32)j : 0u;
}

// after the first sort: isa[suffix] = start of its run; flag elements of runs longer than one
__global__ void k_first_ranks(const u32* sa, const u32* run_start, u64 n, u32* isa, u8* ambiguous) {
  u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  u32 rs = run_start[j];
  isa[sa[j]] = rs;
  bool single = (rs == j) && (j + 1 == n || run_start[j + 1] == j + 1);
  ambiguous[j] = !single;
}
__global__ void k_iota(u32* a, u64 n) {
  u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) a[j] = (u32)j;
}

// doubling round: key = (group start << 32) | (rank of suffix s+h, +1; 0 past the end)
__global__ void k_round_keys(const u32* act_pos, u64 na, const u32* sa, const u32* isa, u64 n, u64 h, u64* key, u32* val) {
  u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= na) return;
  u32 s = sa[act_pos[j]];
  u64 g = isa[s];
  u64 t = (u64)s + h;
  u64 sub = t < n ? (u64)isa[t] + 1 : 0;
  key[j] = (g << 32) | sub;
  val[j] = s;
}
__global__ void k_round_apply(const u64* key, const u32* val, const u32* run_start, const u32* grp_start, u64 na, u32* sa,
                              u32* isa, u32* new_pos, u8* still) {
  u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= na) return;
  u32 g = (u32)(key[j] >> 32);
  u32 gs = grp_start[j], rs = run_start[j];
  u32 p = g + (u32)(j - gs);
  sa[p] = val[j];
  isa[val[j]] = g + (rs - gs);
  new_pos[j] = p;
  bool single = (rs == j) && (j + 1 == na || run_start[j + 1] == j + 1);
  still[j] = !single;
}

__global__ void k_bwt(const u8* text, const u32* sa, u64 n, u8* bwt) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 s = sa[i];
  bwt[i] = s ? text[s - 1] : 0;  // S[n-1] is the sentinel
}

struct Member {
  u8 m[256];  // 0 = symbol does not pass this node, 1 = passes with bit 0, 2 = passes with bit 1
};
// one entry per wavefront: how many of its 64 symbols pass the node
__global__ void k_node_count(const u8* bwt, u64 n, Member mb, u32* wave_cnt, u64 nwaves) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  bool mem = i < n && mb.m[bwt[i]] != 0;
  unsigned long long M = __ballot(mem);
  if ((threadIdx.x & 63) == 0 && (i >> 6) < nwaves) wave_cnt[i >> 6] = (u32)__popcll(M);
}
__global__ void k_node_bits(const u8* bwt, u64 n, Member mb, const u32* wave_off, u64 node_pos, unsigned long long* bv) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u32 lane = threadIdx.x & 63;
  u32 kind = i < n ? mb.m[bwt[i]] : 0;
  unsigned long long M = __ballot(kind != 0);
  if (!M) return;
  u64 base = node_pos + wave_off[i >> 6];
  u32 r = (u32)__popcll(M & ((1ULL << lane) - 1));
  u64 at = base + r;
  u64 w0 = base >> 6;
  unsigned long long c0 = 0, c1 = 0;
  if (kind == 2) {
    if ((at >> 6) == w0) c0 = 1ULL << (at & 63);
    else c1 = 1ULL << (at & 63);
  }
  for (int o = 32; o > 0; o >>= 1) {
    c0 |= __shfl_xor(c0, o);
    c1 |= __shfl_xor(c1, o);
  }
  if (lane == 0) {
    if (c0) atomicOr(&bv[w0], c0);
    if (c1) atomicOr(&bv[w0 + 1], c1);
  }
}

// rank_support_v<1,1>: superblock k = words [8k, 8k+8): bb[2k] = ones before it, bb[2k+1] = seven 9-bit running counts.
// Entries for word boundaries beyond the vector stay 0, exactly as sdsl's constructor leaves them.
__global__ void k_sb_pop(const u64* bv, u64 nwords, u64 nsb, u64* sb_pop) {
  u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nsb) return;
  u64 s = 0;
  for (u32 j = 0; j < 8; ++j) {
    u64 w = 8 * k + j;
    if (w < nwords) s += (u64)__popcll(bv[w]);
  }
  sb_pop[k] = s;
}
__global__ void k_rank_blocks(const u64* bv, u64 nwords, u64 nsb, const u64* sb_before, u64* bb) {
  u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nsb) return;
  u64 rel = 0, run = 0;
  for (u32 j = 1; j <= 7; ++j) {
    u64 w = 8 * k + j - 1;
    if (w < nwords) run += (u64)__popcll(bv[w]);
    if (8 * k + j <= nwords) rel |= run << (63 - 9 * j);
  }
  bb[2 * k] = sb_before[k];
  bb[2 * k + 1] = rel;
}
__global__ void k_gather_stride(const u32* src, u64 n, u32 stride, u32* dst) {
  u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k * stride < n) dst[k] = src[k * stride];
}
__global__ void k_isa_samples(const u32* sa, u64 n, u32* dst) {  // dst[p/64] = i for every SA[i] = p with p % 64 == 0
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 p = sa[i];
  if ((p & 63) == 0) dst[p >> 6] = (u32)i;
}

struct MaxOp {
  __device__ __host__ u32 operator()(u32 a, u32 b) const { return a > b ? a : b; }
};

static u32 grid_for(u64 n, u32 tb = 256) { return (u32)((n + tb - 1) / tb); }

static int max_scan(Scratch& tmp, const u32* in, u32* out, u64 n, hipStream_t st) {
  size_t bytes = 0;
  DG_HIP(rocprim::inclusive_scan(nullptr, bytes, in, out, (size_t)n, MaxOp(), st));
  DG_TRY(tmp.need(bytes));
  DG_HIP(rocprim::inclusive_scan(tmp.p, bytes, in, out, (size_t)n, MaxOp(), st));
  return DG_OK;
}

static int sort_pairs(Scratch& tmp, u64* kin, u64* kout, u32* vin, u32* vout, u64 n, u32 bits, hipStream_t st) {
  size_t bytes = 0;
  DG_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)n, 0u, bits, st));
  DG_TRY(tmp.need(bytes));
  DG_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, (size_t)n, 0u, bits, st));
  return DG_OK;
}

// compaction of 32-bit values by byte flags; returns the kept count
static int compact(Scratch& tmp, const u32* in, const u8* flags, u32* out, u64 n, u64* kept, u64* d_count, hipStream_t st) {
  size_t bytes = 0;
  DG_HIP(rocprim::select(nullptr, bytes, in, flags, out, d_count, (size_t)n, st));
  DG_TRY(tmp.need(bytes));
  DG_HIP(rocprim::select(tmp.p, bytes, in, flags, out, d_count, (size_t)n, st));
  DG_HIP(hipMemcpyAsync(kept, d_count, 8, hipMemcpyDeviceToHost, st));
  DG_HIP(hipStreamSynchronize(st));
  return DG_OK;
}

static int build_impl(const u8* d_text, u64 len, const char* out_path) {
  if (len == 0) return fail(DG_EINVAL, "dg_index_build: empty text");
  const u64 n = len + 1;
  if (n > 0xFFFFFFFFULL) return fail(DG_ELIMIT, "text of %llu symbols exceeds 32-bit suffix-array entries", (unsigned long long)n);
  hipStream_t st = nullptr;
  DG_HIP(hipStreamCreate(&st));
  struct StreamGuard {
    hipStream_t s;
    ~StreamGuard() { (void)hipStreamDestroy(s); }
  } guard{st};
  Scratch tmp;
  const u32 TB = 256;

  // ---- 1. alphabet
  DArr<unsigned long long> d_hist;
  DG_TRY(d_hist.alloc(256));
  DG_HIP(hipMemsetAsync(d_hist.p, 0, 256 * 8, st));
  hipLaunchKernelGGL(k_histogram, dim3(2048), dim3(TB), 0, st, d_text, len, d_hist.p);
  u64 freq[256];
  DG_HIP(hipMemcpyAsync(freq, d_hist.p, 256 * 8, hipMemcpyDeviceToHost, st));
  DG_HIP(hipStreamSynchronize(st));
  if (freq[0]) return fail(DG_EINVAL, "text contains %llu NUL bytes", (unsigned long long)freq[0]);
  freq[0] = 1;  // the sentinel sdsl appends
  CodeMap cm;
  std::memset(&cm, 0, sizeof cm);
  u32 sigma = 0;
  std::vector<u8> comp2char;
  std::vector<u64> C(1, 0);
  u8 char2comp[256];
  std::memset(char2comp, 0, 256);
  for (int c = 0; c < 256; ++c)
    if (freq[c]) {
      cm.code[c] = (u8)sigma;
      char2comp[c] = (u8)sigma;
      comp2char.push_back((u8)c);
      C.push_back(C.back() + freq[c]);
      ++sigma;
    }
  u32 bps = 1;
  while ((1u << bps) < sigma) ++bps;
  const u32 K = 64 / bps;

  // ---- 2. suffix array
  DArr<u32> sa, isa;
  DG_TRY(sa.alloc(n));
  DG_TRY(isa.alloc(n));
  u64 na = 0;
  DArr<u32> act;  // SA positions still ambiguous
  DArr<unsigned long long> d_count;
  DG_TRY(d_count.alloc(1));
  {
    DArr<u64> k0, k1;
    DArr<u32> v0, heads;
    DArr<u8> amb;
    DG_TRY(k0.alloc(n));
    DG_TRY(k1.alloc(n));
    DG_TRY(v0.alloc(n));
    hipLaunchKernelGGL(k_init_keys, dim3(grid_for(n)), dim3(TB), 0, st, d_text, n, cm, bps, K, k0.p, v0.p);
    DG_TRY(sort_pairs(tmp, k0.p, k1.p, v0.p, sa.p, n, K * bps, st));
    k0.release();
    v0.release();
    DArr<u32> starts;
    DG_TRY(heads.alloc(n));
    DG_TRY(starts.alloc(n));
    DG_TRY(amb.alloc(n));
    hipLaunchKernelGGL(k_run_heads, dim3(grid_for(n)), dim3(TB), 0, st, k1.p, n, heads.p);
    k1.release();
    DG_TRY(max_scan(tmp, heads.p, starts.p, n, st));
    heads.release();
    hipLaunchKernelGGL(k_first_ranks, dim3(grid_for(n)), dim3(TB), 0, st, sa.p, starts.p, n, isa.p, amb.p);
    // ambiguous positions, ascending
    DArr<u32> iota;
    DG_TRY(iota.alloc(n));
    DG_TRY(act.alloc(n));
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(TB), 0, st, iota.p, n);
    DG_TRY(compact(tmp, iota.p, amb.p, act.p, n, &na, (u64*)d_count.p, st));
  }
  DG_HIP(hipGetLastError());
  for (u64 h = K; na > 0; h <<= 1) {
    if (h > (n << 1)) return fail(DG_EHIP, "internal: prefix doubling did not converge");
    DArr<u64> k0, k1;
    DArr<u32> v0, v1, rs, gs, rh, gh, npos, nact;
    DArr<u8> still;
    DG_TRY(k0.alloc(na));
    DG_TRY(k1.alloc(na));
    DG_TRY(v0.alloc(na));
    DG_TRY(v1.alloc(na));
    DG_TRY(rs.alloc(na));
    DG_TRY(gs.alloc(na));
    DG_TRY(rh.alloc(na));
    DG_TRY(gh.alloc(na));
    DG_TRY(npos.alloc(na));
    DG_TRY(nact.alloc(na));
    DG_TRY(still.alloc(na));
    hipLaunchKernelGGL(k_round_keys, dim3(grid_for(na)), dim3(TB), 0, st, act.p, na, sa.p, isa.p, n, h, k0.p, v0.p);
    DG_TRY(sort_pairs(tmp, k0.p, k1.p, v0.p, v1.p, na, 64, st));
    hipLaunchKernelGGL(k_run_heads, dim3(grid_for(na)), dim3(TB), 0, st, k1.p, na, rh.p);
    hipLaunchKernelGGL(k_group_heads, dim3(grid_for(na)), dim3(TB), 0, st, k1.p, na, gh.p);
    DG_TRY(max_scan(tmp, rh.p, rs.p, na, st));
    DG_TRY(max_scan(tmp, gh.p, gs.p, na, st));
    hipLaunchKernelGGL(k_round_apply, dim3(grid_for(na)), dim3(TB), 0, st, k1.p, v1.p, rs.p, gs.p, na, sa.p, isa.p, npos.p,
                       still.p);
    u64 kept = 0;
    DG_TRY(compact(tmp, npos.p, still.p, nact.p, na, &kept, (u64*)d_count.p, st));
    DG_HIP(hipMemcpyAsync(act.p, nact.p, kept * 4, hipMemcpyDeviceToDevice, st));
    DG_HIP(hipStreamSynchronize(st));
    na = kept;
  }
  act.release();
  DG_HIP(hipGetLastError());

  // ---- 3. BWT, wavelet tree bits, rank blocks, samples
  HuffTree tree = build_huffman(freq);
  const u64 bv_bits = tree.bv_bits, nwords = (bv_bits + 63) >> 6;
  DArr<u8> bwt;
  DG_TRY(bwt.alloc(n));
  hipLaunchKernelGGL(k_bwt, dim3(grid_for(n)), dim3(TB), 0, st, d_text, sa.p, n, bwt.p);
  DArr<u64> bv;
  DG_TRY(bv.alloc(nwords + 1));
  DG_HIP(hipMemsetAsync(bv.p, 0, (nwords + 1) * 8, st));
  {
    const u64 nwaves = (n + 63) >> 6;
    DArr<u32> wcnt, woff;
    DG_TRY(wcnt.alloc(nwaves));
    DG_TRY(woff.alloc(nwaves));
    for (u32 v = 0; v < tree.nodes.size(); ++v) {
      if (tree.nodes[v].child[0] == 0xFFFF) continue;
      Member mb;
      std::memset(&mb, 0, sizeof mb);
      for (int c = 0; c < 256; ++c) {
        if (tree.c_to_leaf[c] == 0xFFFF) continue;
        u64 p = tree.path[c];
        u32 plen = (u32)(p >> 56), node = 0;
        for (u32 l = 0; l < plen; ++l, p >>= 1) {
          if (node == v) mb.m[c] = (p & 1) ? 2 : 1;
          node = tree.nodes[node].child[p & 1];
        }
      }
      hipLaunchKernelGGL(k_node_count, dim3(grid_for(nwaves * 64)), dim3(TB), 0, st, bwt.p, n, mb, wcnt.p, nwaves);
      size_t bytes = 0;
      DG_HIP(rocprim::exclusive_scan(nullptr, bytes, wcnt.p, woff.p, 0u, (size_t)nwaves, rocprim::plus<u32>(), st));
      DG_TRY(tmp.need(bytes));
      DG_HIP(rocprim::exclusive_scan(tmp.p, bytes, wcnt.p, woff.p, 0u, (size_t)nwaves, rocprim::plus<u32>(), st));
      hipLaunchKernelGGL(k_node_bits, dim3(grid_for(nwaves * 64)), dim3(TB), 0, st, bwt.p, n, mb, woff.p, tree.nodes[v].bv_pos,
                         (unsigned long long*)bv.p);
    }
  }
  bwt.release();
  const u64 nsb = (nwords >> 3) + 1;  // ((capacity >> 9) + 1) superblocks
  DArr<u64> sbp, sbb, bb;
  DG_TRY(sbp.alloc(nsb));
  DG_TRY(sbb.alloc(nsb));
  DG_TRY(bb.alloc(2 * nsb));
  hipLaunchKernelGGL(k_sb_pop, dim3(grid_for(nsb)), dim3(TB), 0, st, bv.p, nwords, nsb, sbp.p);
  {
    size_t bytes = 0;
    DG_HIP(rocprim::exclusive_scan(nullptr, bytes, sbp.p, sbb.p, (u64)0, (size_t)nsb, rocprim::plus<u64>(), st));
    DG_TRY(tmp.need(bytes));
    DG_HIP(rocprim::exclusive_scan(tmp.p, bytes, sbp.p, sbb.p, (u64)0, (size_t)nsb, rocprim::plus<u64>(), st));
  }
  hipLaunchKernelGGL(k_rank_blocks, dim3(grid_for(nsb)), dim3(TB), 0, st, bv.p, nwords, nsb, sbb.p, bb.p);
  const u64 n_sa_s = (n + 31) / 32, n_isa_s = (n - 1) / 64 + 1;
  DArr<u32> sas, isas;
  DG_TRY(sas.alloc(n_sa_s));
  DG_TRY(isas.alloc(n_isa_s));
  hipLaunchKernelGGL(k_gather_stride, dim3(grid_for(n_sa_s)), dim3(TB), 0, st, sa.p, n, 32u, sas.p);
  hipLaunchKernelGGL(k_isa_samples, dim3(grid_for(n)), dim3(TB), 0, st, sa.p, n, isas.p);
  // ---- 4. host side
  std::vector<u64> h_bv(nwords), h_bb(2 * nsb);
  std::vector<u32> h_sas(n_sa_s), h_isas(n_isa_s);
  DG_HIP(hipMemcpyAsync(h_bv.data(), bv.p, nwords * 8, hipMemcpyDeviceToHost, st));
  DG_HIP(hipMemcpyAsync(h_bb.data(), bb.p, 2 * nsb * 8, hipMemcpyDeviceToHost, st));
  DG_HIP(hipMemcpyAsync(h_sas.data(), sas.p, n_sa_s * 4, hipMemcpyDeviceToHost, st));
  DG_HIP(hipMemcpyAsync(h_isas.data(), isas.p, n_isa_s * 4, hipMemcpyDeviceToHost, st));
  DG_HIP(hipStreamSynchronize(st));
  DG_HIP(hipGetLastError());
  sa.release();
  isa.release();
  // init_node_ranks: inner nodes carry rank1(bv_pos)
  auto rank1 = [&](u64 idx) -> u64 {
    const u64* p = h_bb.data() + ((idx >> 8) & ~1ULL);
    u64 r = p[0] + ((p[1] >> (63 - 9 * ((idx & 0x1FF) >> 6))) & 0x1FF);
    if (idx & 63) r += (u64)__builtin_popcountll(h_bv[idx >> 6] & ((1ULL << (idx & 63)) - 1));
    return r;
  };
  for (auto& nd : tree.nodes)
    if (nd.child[0] != 0xFFFF) nd.bv_pos_rank = rank1(nd.bv_pos);

  FileOut o;
  if (!o.open(out_path)) return fail(DG_EIO, "cannot create %s", out_path);
  u64 hash = kPlaceholderClassHash;
  if (const char* hs = std::getenv("DICEY_FM9_HASH")) hash = std::strtoull(hs, nullptr, 0);
  o.u64v(hash);
  o.u64v(n);
  o.u64v(sigma);
  o.u64v(bv_bits);
  o.raw(h_bv.data(), nwords * 8);
  o.u64v(2 * nsb * 64);
  o.raw(h_bb.data(), 2 * nsb * 8);
  write_select_support(o, h_bv.data(), bv_bits, true);
  write_select_support(o, h_bv.data(), bv_bits, false);
  o.u64v((u64)tree.nodes.size());
  for (const auto& nd : tree.nodes) {
    o.u64v(nd.bv_pos);
    o.u64v(nd.bv_pos_rank);
    o.u16v(nd.parent);
    o.u16v(nd.child[0]);
    o.u16v(nd.child[1]);
  }
  o.raw(tree.c_to_leaf, sizeof tree.c_to_leaf);
  o.raw(tree.path, sizeof tree.path);
  const u8 wd = (u8)(bits_hi(n) + 1);
  auto write_samples = [&](const std::vector<u32>& v) {
    o.u64v((u64)v.size() * wd);
    o.u8v(wd);
    if (wd == 32) {
      o.raw(v.data(), v.size() * 4);
      if (v.size() & 1) {
        u32 z = 0;
        o.raw(&z, 4);
      }
    } else {
      PackedVec pv;
      pv.init(v.size(), wd);
      for (u64 i = 0; i < v.size(); ++i) pv.set(i, v[i]);
      o.raw(pv.w.data(), pv.w.size() * 8);
    }
  };
  write_samples(h_sas);
  write_samples(h_isas);
  o.u64v(2048);
  o.raw(char2comp, 256);
  o.u64v((u64)sigma * 8);
  {
    std::vector<u8> padded((sigma * 8 + 63) / 64 * 8, 0);
    std::memcpy(padded.data(), comp2char.data(), sigma);
    o.raw(padded.data(), padded.size());
  }
  o.u64v((u64)C.size() * 64);
  o.raw(C.data(), C.size() * 8);
  o.u16v((u16)sigma);
  if (!o.close()) return fail(DG_EIO, "short write to %s", out_path);
  return DG_OK;
}

}  // namespace

// Used by dg_index_open (index.hip): the suffix array as the inverse of the inverse suffix array.  isa[p] = rank of the
// suffix at p; sorting the pairs (isa[p], p) by their key leaves the positions in rank order, i.e. SA.  Four 8-bit radix
// passes stream the 3.1 G pairs at memory speed, where writing sa[isa[p]] = p directly is 3.1 G random 4-byte stores
// (measured: ~2.6 G stores/s, 1.3 s).  `isa` is clobbered.
int derive_sa_by_sort(hipStream_t st, u32* isa, u64 n, u32* sa) {
  u32* keys_out = nullptr;
  DG_HIP(big_alloc((void**)&keys_out, n * 4 + 64, st));
  Scratch tmp;
  u32 bits = 1;
  while (bits < 32 && (1ULL << bits) < n) ++bits;
  size_t bytes = 0;
  rocprim::counting_iterator<u32> pos(0);
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, isa, keys_out, pos, sa, (size_t)n, 0u, bits, st);
  int rc = DG_OK;
  if (e == hipSuccess) rc = tmp.need(bytes);
  if (e == hipSuccess && rc == DG_OK) e = rocprim::radix_sort_pairs(tmp.p, bytes, isa, keys_out, pos, sa, (size_t)n, 0u, bits, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  big_free(keys_out, st);
  if (rc != DG_OK) return rc;
  if (e != hipSuccess) return fail(DG_EHIP, "suffix array by sort: %s", hipGetErrorString(e));
  return DG_OK;
}
}  // namespace dg

using namespace dg;

extern "C" {

int dg_index_build_device(const void* d_text, uint64_t len, int device, const char* out_fm9_path) {
  if (!d_text || !out_fm9_path) return fail(DG_EINVAL, "dg_index_build_device: null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DG_ENODEV, "no HIP device available");
  if (device < 0 || device >= ndev) return fail(DG_EINVAL, "device %d out of range (have %d)", device, ndev);
  DG_HIP(hipSetDevice(device));
  return build_impl((const u8*)d_text, len, out_fm9_path);
}

int dg_index_build(const uint8_t* text, uint64_t len, int device, const char* out_fm9_path) {
  if (!text || !out_fm9_path) return fail(DG_EINVAL, "dg_index_build: null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DG_ENODEV, "no HIP device available");
  if (device < 0 || device >= ndev) return fail(DG_EINVAL, "device %d out of range (have %d)", device, ndev);
  DG_HIP(hipSetDevice(device));
  void* d = nullptr;
  DG_HIP(hipMalloc(&d, len ? len : 1));
  hipError_t e = hipMemcpy(d, text, len, hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? build_impl((const u8*)d, len, out_fm9_path) : fail(DG_EHIP, "upload failed: %s", hipGetErrorString(e));
  (void)hipFree(d);
  return rc;
}

}  // extern "C"
