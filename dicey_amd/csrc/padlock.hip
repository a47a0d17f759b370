// `dicey padlock`: everything the reference computes per exon position before it decides (src/padlock.h:321-428), for a
// batch of exons in one call.  The reference calls thal() up to three times, sdsl::count four times and count() over two
// neighbourhoods per position, one position after the other; here the positions of all exons go through the GPU in
// stages — thal of every arm window that passes the GC filter, thal of every probe whose two arms pass, exact and
// neighbourhood occurrence counts of the arms of probes inside the Tm window — and the caller replays the reference's
// decision sequence on the returned arrays (the values do not depend on which positions the reference skips).
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"
#include "iupac.hpp"
#include "thal_internal.hpp"

using namespace dg;

namespace {

// Host threads for the per-position loops (r05: 2.7 M positions of 1 000 genes spent 0.12 of a 0.35 s step in single-threaded loops
// around the GPU stages): DICEY_HOST_THREADS, else up to 16.
unsigned scan_threads() {
  static const unsigned n = [] {
    const char* e = std::getenv("DICEY_HOST_THREADS");
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const long v = e ? std::atol(e) : (long)std::min(16u, hw);
    return (unsigned)std::max(1L, std::min(v, 256L));
  }();
  return n;
}
// f(chunk, begin, end) over [0, n) in nt contiguous chunks with the given boundaries (cut[0] = 0 .. cut[nt] = n)
template <class F>
void run_chunks(const std::vector<size_t>& cut, F f) {
  const size_t nt = cut.size() - 1;
  if (nt <= 1) {
    f((size_t)0, cut.front(), cut.back());
    return;
  }
  // an empty chunk (two equal boundaries: one exon longer than a chunk's share of the positions) still gets its call — the
  // two-pass stages write a count per chunk and a chunk that kept the previous stage's value would shift every later one
  std::vector<std::thread> th;
  th.reserve(nt);
  for (size_t t = 0; t < nt; ++t) {
    if (cut[t] < cut[t + 1]) th.emplace_back([&f, &cut, t] { f(t, cut[t], cut[t + 1]); });
    else f(t, cut[t], cut[t]);
  }
  for (auto& x : th) x.join();
}
std::vector<size_t> even_cuts(size_t n, unsigned nt, size_t grain) {
  const size_t parts = std::max<size_t>(1, std::min<size_t>(nt, n / std::max<size_t>(1, grain)));
  std::vector<size_t> cut(parts + 1);
  for (size_t t = 0; t <= parts; ++t) cut[t] = n * t / parts;
  return cut;
}

// r06: the six per-position arrays of a result are ONE block that goes back to a small pool when the result is freed: a caller that
// scans batch after batch (the binary over a gene list, bench.py) gets memory its threads have touched before — 128 MB of
// first-touch page faults per 2.7 M positions were a tenth of the step
struct ResultBox {
  dg_padlock_result pub;  // (first member: the public pointer is the box's)
  void* block = nullptr;
  size_t cap = 0;         // positions the block holds
};
std::mutex g_pool_mu;
std::vector<std::pair<void*, size_t>> g_pool;  // (block, positions); at most two wait here
void* pool_take(size_t npos, size_t& cap) {
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); ++i)
      if (g_pool[i].second >= npos && g_pool[i].second <= 2 * npos + 4096) {
        void* p = g_pool[i].first;
        cap = g_pool[i].second;
        g_pool.erase(g_pool.begin() + (long)i);
        return p;
      }
  }
  cap = npos + npos / 16 + 64;
  return std::malloc(cap * 6 * 8);
}
void pool_give(void* p, size_t cap) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (g_pool.size() >= 2) {
    std::free(g_pool.front().first);
    g_pool.erase(g_pool.begin());
  }
  g_pool.emplace_back(p, cap);
}

}  // namespace

extern "C" {

void dg_padlock_result_free(dg_padlock_result* r) {
  if (!r) return;
  ResultBox* box = reinterpret_cast<ResultBox*>(r);
  delete[] r->pos_off;
  pool_give(box->block, box->cap);
  delete box;
}

int dg_padlock_scan(dg_index* ix, dg_thal* th, const dg_padlock_params* p, const uint8_t* exons, const uint64_t* exon_off, size_t nexons,
                    dg_padlock_result** out) {
  if (!ix || !th || !p || !exon_off || !out || (!exons && nexons && exon_off[nexons])) return fail(DG_EINVAL, "dg_padlock_scan: null argument");
  *out = nullptr;
  const uint64_t L = p->armlen, T = 2 * L;
  if (L == 0) return fail(DG_EINVAL, "dg_padlock_scan: arm length 0");
  ResultBox* box = new ResultBox;
  dg_padlock_result* R = &box->pub;
  std::memset(R, 0, sizeof *R);
  R->nexons = nexons;
  R->pos_off = new uint64_t[nexons + 1];
  uint64_t npos = 0;
  for (size_t e = 0; e < nexons; ++e) {
    if (exon_off[e + 1] < exon_off[e]) {
      dg_padlock_result_free(R);
      return fail(DG_EINVAL, "dg_padlock_scan: exon_off must be non-decreasing");
    }
    R->pos_off[e] = npos;
    const uint64_t len = exon_off[e + 1] - exon_off[e];
    if (len >= T) npos += len - L + 1;  // shorter exons carry no probe (padlock.h:318)
  }
  R->pos_off[nexons] = npos;
  R->npos = npos;
  box->block = pool_take(npos ? npos : 1, box->cap);
  if (!box->block) {
    delete[] R->pos_off;
    delete box;
    return fail(DG_ENOMEM, "dg_padlock_scan: out of host memory (%llu positions)", (unsigned long long)npos);
  }
  {
    double* d = static_cast<double*>(box->block);
    const size_t c = box->cap;
    R->arm_gc = d;
    R->arm_tm = d + c;
    R->probe_gc = d + 2 * c;
    R->probe_tm = d + 3 * c;
    R->arm_count = reinterpret_cast<int64_t*>(d + 4 * c);
    R->arm_nbcount = reinterpret_cast<int64_t*>(d + 5 * c);
  }
  static const bool timing = std::getenv("DICEY_TIMING") != nullptr;  // debugging aid: host-side phase times on stderr
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "dg_padlock_scan: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  auto fail_with = [&](int rc) {
    dg_padlock_result_free(R);
    return rc;
  };
  const double minGC = p->gc_min, maxGC = p->gc_max;
  // reverse complement of a window = what rexonseq.substr(exonlen - len - k, len) is in the reference
  auto revcomp_into = [&](std::string& buf, const uint8_t* s, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) buf.push_back(complement_iupac((char)s[n - 1 - i]));
  };
  const uint64_t nbytes = nexons ? exon_off[nexons] : 0;
  // thal(window, its reverse complement) for windows of ONE length at the given byte offsets
  auto thal_windows = [&](const std::vector<uint64_t>& wo, uint64_t wlen, std::vector<double>& temps) -> int {
    temps.assign(wo.size(), 0.0);
    if (wo.empty()) return DG_OK;
    if (wlen <= kSelfWindowMax) {  // pairs formed on the device from the exon bytes
      return thal_self_windows(th, exons, nbytes, wo.data(), nullptr, wo.size(), temps.data(), (uint32_t)wlen);
    }
    std::string buf;
    std::vector<uint64_t> off(1, 0);
    for (uint64_t w : wo) {
      buf.append((const char*)exons + w, wlen);
      off.push_back(buf.size());
      revcomp_into(buf, exons + w, wlen);
      off.push_back(buf.size());
    }
    return dg_thal_batch(th, (const uint8_t*)buf.data(), off.data(), wo.size(), temps.data(), nullptr, nullptr);
  };
  // gccontent() of any window from running counts: G/C so far, and N/n so far (a window with an N scores -1)
  // (r06: buffers this thread keeps between calls; the counts restart at every exon — a window never leaves its exon — so that the
  //  exons' chunks can be counted by the host threads that scan them: entry i + e + 1 = characters of exon e in front of byte i)
  static thread_local std::vector<uint32_t> gcs_buf, nns_buf;
  if (gcs_buf.size() < nbytes + nexons + 1) {
    gcs_buf.resize(nbytes + nexons + 1);
    nns_buf.resize(nbytes + nexons + 1);
  }
  // (plain pointers: the lambdas below run on other threads, where a thread_local NAME would mean that thread's own, empty vector)
  uint32_t* const gcs_p = gcs_buf.data();
  uint32_t* const nns_p = nns_buf.data();
  auto gc_of = [&](size_t e, uint64_t at, uint64_t n) -> double {  // window [at, at + n) of exon e (byte offsets)
    const uint32_t* g = gcs_p + e;
    const uint32_t* nn = nns_p + e;
    if (nn[at + n] != nn[at]) return -1;
    return (double)(g[at + n] - g[at]) / (double)n;
  };
  // exons in contiguous chunks of about equal numbers of positions, one host thread each (r05); a stage is two passes over a
  // chunk — count the windows it sends to the GPU, then write them at the chunk's place of the common list: the lists come out in
  // exon order whatever the number of threads
  std::vector<size_t> ecut(1, 0);
  {
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(scan_threads(), npos / 65536));
    for (unsigned t = 1; t < nt; ++t) {
      const uint64_t want = npos * t / nt;
      const size_t e = (size_t)(std::upper_bound(R->pos_off, R->pos_off + nexons + 1, want) - R->pos_off) - 1;
      ecut.push_back(std::max(ecut.back(), std::min(e, nexons)));
    }
    ecut.push_back(nexons);
  }
  const size_t nchunks = ecut.size() - 1;
  std::vector<uint64_t> cnt(nchunks + 1, 0);
  auto place = [&](std::vector<uint64_t>& wo, std::vector<uint64_t>& where) {  // counts per chunk -> first slots, lists sized
    uint64_t total = 0;
    for (size_t c = 0; c < nchunks; ++c) {
      const uint64_t n = cnt[c];
      cnt[c] = total;
      total += n;
    }
    cnt[nchunks] = total;
    wo.resize(total);
    where.resize(total);
  };
  // stage 1: GC of every arm window; thal(arm, reverse complement) where the GC filter lets it through (padlock.h:323-345)
  static thread_local std::vector<uint64_t> wo_buf, where_buf;  // (kept between calls: 24 MB of value-initialisation and page faults per stage otherwise)
  std::vector<uint64_t>&wo = wo_buf, &where = where_buf;
  run_chunks(ecut, [&](size_t c, size_t e0, size_t e1) {
    uint64_t n = 0;
    for (size_t e = e0; e < e1; ++e) {
      const uint64_t b0 = exon_off[e], len = exon_off[e + 1] - b0;
      if (len < T) continue;
      {  // the exon's running counts (G/C, N) — see gc_of
        uint32_t* g = gcs_p + e + b0;
        uint32_t* nn = nns_p + e + b0;
        uint32_t cg = 0, cn = 0;
        g[0] = nn[0] = 0;
        for (uint64_t i = 0; i < len; ++i) {
          const char ch = (char)exons[b0 + i];
          cg += (ch == 'C' || ch == 'G' || ch == 'c' || ch == 'g');
          cn += (ch == 'N' || ch == 'n');
          g[i + 1] = cg;
          nn[i + 1] = cn;
        }
      }
      for (uint64_t q = 0; q + L <= len; ++q) {
        const uint64_t at = R->pos_off[e] + q;
        const double gc = gc_of(e, b0 + q, L);
        R->arm_gc[at] = gc;
        R->arm_tm[at] = R->probe_tm[at] = DG_PADLOCK_NOT_COMPUTED;
        R->probe_gc[at] = 0;
        R->arm_count[at] = R->arm_nbcount[at] = -1;
        n += !(gc < minGC || gc > maxGC);
      }
    }
    cnt[c] = n;
  });
  place(wo, where);
  run_chunks(ecut, [&](size_t c, size_t e0, size_t e1) {
    uint64_t k = cnt[c];
    for (size_t e = e0; e < e1; ++e) {
      const uint64_t b0 = exon_off[e], len = exon_off[e + 1] - b0;
      if (len < T) continue;
      for (uint64_t q = 0; q + L <= len; ++q) {
        const uint64_t at = R->pos_off[e] + q;
        if (R->arm_gc[at] < minGC || R->arm_gc[at] > maxGC) continue;
        wo[k] = b0 + q;
        where[k++] = at;
      }
    }
  });
  lap("arm GC + window list");
  static thread_local std::vector<double> temps_buf;
  std::vector<double>& temps = temps_buf;
  int rc = thal_windows(wo, L, temps);
  lap("arm thal (pack + GPU)");
  if (rc != DG_OK) return fail_with(rc);
  const uint64_t n_arm_thal = where.size();
  run_chunks(even_cuts(where.size(), scan_threads(), 65536), [&](size_t, size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; ++i) R->arm_tm[where[i]] = temps[i];
  });
  // stage 2: probes whose two arms pass GC, Tm ceiling and Tm difference (padlock.h:327-362)
  auto arm_ok = [&](uint64_t at) {
    const double gc = R->arm_gc[at];
    if (gc < minGC || gc > maxGC) return false;
    return !(R->arm_tm[at] > 93 + gc - 675.0 / (double)p->armlen);  // a refused thal (-999999) is the caller's error path
  };
  auto probe_goes = [&](uint64_t at) {  // (probe_gc[at] is written by the first pass)
    if (!arm_ok(at) || !arm_ok(at + L)) return false;
    if (std::abs(R->arm_tm[at] - R->arm_tm[at + L]) > (double)p->tmdiff) return false;
    return !(R->probe_gc[at] < minGC || R->probe_gc[at] > maxGC);
  };
  std::fill(cnt.begin(), cnt.end(), 0);  // (stage 1 left its chunks' first slots here)
  run_chunks(ecut, [&](size_t c, size_t e0, size_t e1) {
    uint64_t n = 0;
    for (size_t e = e0; e < e1; ++e) {
      const uint64_t b0 = exon_off[e], len = exon_off[e + 1] - b0;
      if (len < T) continue;
      for (uint64_t k = 0; k + T <= len; ++k) {
        const uint64_t at = R->pos_off[e] + k;
        R->probe_gc[at] = gc_of(e, b0 + k, T);
        n += probe_goes(at);
      }
    }
    cnt[c] = n;
  });
  place(wo, where);
  run_chunks(ecut, [&](size_t c, size_t e0, size_t e1) {
    uint64_t j = cnt[c];
    for (size_t e = e0; e < e1; ++e) {
      const uint64_t b0 = exon_off[e], len = exon_off[e + 1] - b0;
      if (len < T) continue;
      for (uint64_t k = 0; k + T <= len; ++k) {
        const uint64_t at = R->pos_off[e] + k;
        if (!probe_goes(at)) continue;
        wo[j] = b0 + k;
        where[j++] = at;
      }
    }
  });
  lap("probe GC + window list");
  rc = thal_windows(wo, T, temps);
  if (rc != DG_OK) return fail_with(rc);
  for (size_t i = 0; i < where.size(); ++i) R->probe_tm[where[i]] = temps[i];
  lap("probe thal (pack + GPU)");
  // stage 3: arms of probes inside the Tm window: exact occurrences on both strands and the neighbourhood totals
  std::vector<std::pair<uint64_t, uint64_t>> arms;  // (byte offset, result slot)
  for (size_t i = 0; i < where.size(); ++i) {
    const uint64_t at = where[i];
    const double lo = 81.5 + R->probe_gc[at] - 675.0 / (double)(2 * p->armlen), tmv = temps[i];
    if (tmv == -999999.0 || tmv < lo || tmv > lo + 10) continue;
    for (uint64_t a : {at, at + L})
      if (R->arm_count[a] == -1) {
        R->arm_count[a] = -2;  // queued
        arms.emplace_back(wo[i] + (a - at), a);
      }
  }
  if (!arms.empty()) {
    // (every record is L bytes: arm and reverse complement side by side, written by the host threads)
    std::string buf(2 * L * arms.size(), '\0');
    std::vector<uint64_t> off(2 * arms.size() + 1);
    const std::vector<size_t> acut = even_cuts(arms.size(), scan_threads(), 16384);
    run_chunks(acut, [&](size_t, size_t i0, size_t i1) {
      for (size_t i = i0; i < i1; ++i) {
        const uint8_t* src = exons + arms[i].first;
        char* dst = &buf[2 * L * i];
        std::memcpy(dst, src, L);
        for (uint64_t k = 0; k < L; ++k) dst[L + k] = complement_iupac((char)src[L - 1 - k]);
        off[2 * i] = 2 * L * i;
        off[2 * i + 1] = 2 * L * i + L;
      }
    });
    off[2 * arms.size()] = 2 * L * arms.size();
    std::vector<uint64_t> cnt2(2 * arms.size());
    rc = dg_count(ix, (const uint8_t*)buf.data(), off.data(), 2 * arms.size(), cnt2.data());
    if (rc != DG_OK) return fail_with(rc);
    for (size_t i = 0; i < arms.size(); ++i) R->arm_count[arms[i].second] = (int64_t)(cnt2[2 * i] + cnt2[2 * i + 1]);
    lap("exact counts");
    if (p->distance > 0) {
      std::string fbuf(L * arms.size(), '\0');
      std::vector<uint64_t> foff(arms.size() + 1);
      run_chunks(acut, [&](size_t, size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
          std::memcpy(&fbuf[L * i], exons + arms[i].first, L);
          foff[i] = L * i;
        }
      });
      foff[arms.size()] = L * arms.size();
      std::vector<uint64_t> fw(arms.size()), rv(arms.size());
      rc = dg_neighborhood_count(ix, p->distance, p->hamming, 10000, (const uint8_t*)fbuf.data(), foff.data(), arms.size(), fw.data(), rv.data());
      if (rc != DG_OK) return fail_with(rc);
      for (size_t i = 0; i < arms.size(); ++i) R->arm_nbcount[arms[i].second] = (int64_t)(fw[i] + rv[i]);
      lap("neighbourhood counts");
    }
  }
  R->n_arm_thal = n_arm_thal;  // (one thal per arm window that passed the GC filter)
  R->n_probe_thal = where.size();
  R->n_arms_counted = arms.size();
  *out = R;
  return DG_OK;
}

}  // extern "C"
