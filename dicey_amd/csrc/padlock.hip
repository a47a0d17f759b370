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
#include <string>
#include <vector>

#include "common.hpp"
#include "iupac.hpp"
#include "thal_internal.hpp"

using namespace dg;

namespace {

}  // namespace

extern "C" {

void dg_padlock_result_free(dg_padlock_result* r) {
  if (!r) return;
  delete[] r->pos_off;
  delete[] r->arm_gc;
  delete[] r->arm_tm;
  delete[] r->probe_gc;
  delete[] r->probe_tm;
  delete[] r->arm_count;
  delete[] r->arm_nbcount;
  delete r;
}

int dg_padlock_scan(dg_index* ix, dg_thal* th, const dg_padlock_params* p, const uint8_t* exons, const uint64_t* exon_off, size_t nexons,
                    dg_padlock_result** out) {
  if (!ix || !th || !p || !exon_off || !out || (!exons && nexons && exon_off[nexons])) return fail(DG_EINVAL, "dg_padlock_scan: null argument");
  *out = nullptr;
  const uint64_t L = p->armlen, T = 2 * L;
  if (L == 0) return fail(DG_EINVAL, "dg_padlock_scan: arm length 0");
  dg_padlock_result* R = new dg_padlock_result;
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
  R->arm_gc = new double[npos ? npos : 1];
  R->arm_tm = new double[npos ? npos : 1];
  R->probe_gc = new double[npos ? npos : 1];
  R->probe_tm = new double[npos ? npos : 1];
  R->arm_count = new int64_t[npos ? npos : 1];
  R->arm_nbcount = new int64_t[npos ? npos : 1];
  for (uint64_t i = 0; i < npos; ++i) {
    R->arm_tm[i] = R->probe_tm[i] = DG_PADLOCK_NOT_COMPUTED;
    R->probe_gc[i] = 0;
    R->arm_count[i] = R->arm_nbcount[i] = -1;
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
  auto thal_pairs = [&](const std::vector<std::pair<uint64_t, uint64_t>>& win /* (byte offset, length) */, std::vector<double>& temps) -> int {
    temps.assign(win.size(), 0.0);
    if (win.empty()) return DG_OK;
    if (win[0].second <= kSelfWindowMax) {  // pairs formed on the device from the exon bytes
      std::vector<uint64_t> wo(win.size());
      std::vector<uint32_t> wl(win.size());
      for (size_t i = 0; i < win.size(); ++i) {
        wo[i] = win[i].first;
        wl[i] = (uint32_t)win[i].second;
      }
      return thal_self_windows(th, exons, nbytes, wo.data(), wl.data(), win.size(), temps.data());
    }
    std::string buf;
    std::vector<uint64_t> off(1, 0);
    for (auto& w : win) {
      buf.append((const char*)exons + w.first, w.second);
      off.push_back(buf.size());
      revcomp_into(buf, exons + w.first, w.second);
      off.push_back(buf.size());
    }
    return dg_thal_batch(th, (const uint8_t*)buf.data(), off.data(), win.size(), temps.data(), nullptr, nullptr);
  };
  // gccontent() of any window from running counts: G/C so far, and N/n so far (a window with an N scores -1)
  std::vector<uint32_t> gcs(nbytes + 1, 0), nns(nbytes + 1, 0);
  for (uint64_t i = 0; i < nbytes; ++i) {
    const char ch = (char)exons[i];
    gcs[i + 1] = gcs[i] + (ch == 'C' || ch == 'G' || ch == 'c' || ch == 'g');
    nns[i + 1] = nns[i] + (ch == 'N' || ch == 'n');
  }
  auto gc_of = [&](uint64_t at, uint64_t n) -> double {
    if (nns[at + n] != nns[at]) return -1;
    return (double)(gcs[at + n] - gcs[at]) / (double)n;
  };
  // stage 1: GC of every arm window; thal(arm, reverse complement) where the GC filter lets it through (padlock.h:323-345)
  std::vector<std::pair<uint64_t, uint64_t>> win;
  std::vector<uint64_t> where;
  for (size_t e = 0; e < nexons; ++e) {
    const uint64_t b0 = exon_off[e], len = exon_off[e + 1] - b0;
    if (len < T) continue;
    for (uint64_t q = 0; q + L <= len; ++q) {
      const uint64_t at = R->pos_off[e] + q;
      R->arm_gc[at] = gc_of(b0 + q, L);
      if (R->arm_gc[at] < minGC || R->arm_gc[at] > maxGC) continue;
      win.emplace_back(b0 + q, L);
      where.push_back(at);
    }
  }
  lap("arm GC + window list");
  std::vector<double> temps;
  int rc = thal_pairs(win, temps);
  lap("arm thal (pack + GPU)");
  if (rc != DG_OK) return fail_with(rc);
  for (size_t i = 0; i < where.size(); ++i) R->arm_tm[where[i]] = temps[i];
  // stage 2: probes whose two arms pass GC, Tm ceiling and Tm difference (padlock.h:327-362)
  win.clear();
  where.clear();
  auto arm_ok = [&](uint64_t at) {
    const double gc = R->arm_gc[at];
    if (gc < minGC || gc > maxGC) return false;
    return !(R->arm_tm[at] > 93 + gc - 675.0 / (double)p->armlen);  // a refused thal (-999999) is the caller's error path
  };
  for (size_t e = 0; e < nexons; ++e) {
    const uint64_t b0 = exon_off[e], len = exon_off[e + 1] - b0;
    if (len < T) continue;
    for (uint64_t k = 0; k + T <= len; ++k) {
      const uint64_t at = R->pos_off[e] + k;
      R->probe_gc[at] = gc_of(b0 + k, T);
      if (!arm_ok(at) || !arm_ok(at + L)) continue;
      if (std::abs(R->arm_tm[at] - R->arm_tm[at + L]) > (double)p->tmdiff) continue;
      if (R->probe_gc[at] < minGC || R->probe_gc[at] > maxGC) continue;
      win.emplace_back(b0 + k, T);
      where.push_back(at);
    }
  }
  lap("probe GC + window list");
  rc = thal_pairs(win, temps);
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
        arms.emplace_back(win[i].first + (a - at), a);
      }
  }
  if (!arms.empty()) {
    std::string buf;
    std::vector<uint64_t> off(1, 0);
    for (auto& a : arms) {
      buf.append((const char*)exons + a.first, L);
      off.push_back(buf.size());
      revcomp_into(buf, exons + a.first, L);
      off.push_back(buf.size());
    }
    std::vector<uint64_t> cnt(2 * arms.size());
    rc = dg_count(ix, (const uint8_t*)buf.data(), off.data(), 2 * arms.size(), cnt.data());
    if (rc != DG_OK) return fail_with(rc);
    for (size_t i = 0; i < arms.size(); ++i) R->arm_count[arms[i].second] = (int64_t)(cnt[2 * i] + cnt[2 * i + 1]);
    lap("exact counts");
    if (p->distance > 0) {
      std::string fbuf;
      std::vector<uint64_t> foff(1, 0);
      for (auto& a : arms) {
        fbuf.append((const char*)exons + a.first, L);
        foff.push_back(fbuf.size());
      }
      std::vector<uint64_t> fw(arms.size()), rv(arms.size());
      rc = dg_neighborhood_count(ix, p->distance, p->hamming, 10000, (const uint8_t*)fbuf.data(), foff.data(), arms.size(), fw.data(), rv.data());
      if (rc != DG_OK) return fail_with(rc);
      for (size_t i = 0; i < arms.size(); ++i) R->arm_nbcount[arms[i].second] = (int64_t)(fw[i] + rv[i]);
      lap("neighbourhood counts");
    }
  }
  R->n_arm_thal = 0;
  for (uint64_t i = 0; i < npos; ++i) R->n_arm_thal += R->arm_tm[i] != DG_PADLOCK_NOT_COMPUTED;
  R->n_probe_thal = where.size();
  R->n_arms_counted = arms.size();
  *out = R;
  return DG_OK;
}

}  // extern "C"
