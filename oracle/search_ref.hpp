// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under dicey_amd/ may include, link or call this.
//
// CPU restatement of `dicey search` (reference src/silica.h:208-651).  The two pieces of arithmetic that live outside
// silica.h are NOT restated but taken from the reference itself, compiled in place into oracle/_ref/ and passed in as
// function pointers: primer3thal::thal() (src/thal.h) and nlohmann::json::dump() of a double (src/jlib/nlohmann).
// Everything in this file follows the silica.h lines it cites; parity of the driver logic itself is unpinned (the
// reference has no tests and its binary cannot be built here).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <set>
#include <string>
#include <vector>

#include "fm9.hpp"
#include "hunt_ref.hpp"

namespace orc {

typedef int (*ThalFn)(const char* oligo1, const char* oligo2, double* temp, int* end1, int* end2);
typedef int (*DumpDoubleFn)(double x, char* out, int cap);

struct SearchParams {  // silica.h:38-68 with the defaults of :214-252
  bool indel = true, pruneprimer = false;
  double cutTemp = 45.0;
  uint32_t maxProdSize = 15000;
  double cutofPen = -1.0, penDiff = 0.6, penMis = 0.4, penLen = 0.001;
  uint32_t kmer = 15, distance = 1, maxNeighborhood = 10000, maxPruneCount = 0;
  uint64_t max_locations = 10000;
  std::string genome, outfile;
};

struct PrimerBind {  // silica.h:69-82
  uint32_t refIndex, pos, primerId;
  bool onFor;
  double temp, perfTemp;
  std::string genome;
  bool operator<(const PrimerBind& b) const { return temp > b.temp; }
};
struct PcrProduct {  // silica.h:84-98
  uint32_t refIndex, leng, forPos, revPos, forId, revId;
  double forTemp, revTemp, penalty;
  bool operator<(const PcrProduct& b) const { return penalty < b.penalty; }
};

struct SearchRun {
  const Csa* fm;
  std::vector<uint32_t> seqlen;
  std::vector<std::string> seqname;
  const std::string* text;  // the index text (upper-case): source of amplicon sequences (faidx_fetch_seq + to_upper)
  ThalFn thal;
  DumpDoubleFn dump_double;
  SearchParams c;

  std::string num(double x) const {
    char b[64];
    dump_double(x, b, 64);
    return b;
  }

  // silica.h:100-187
  std::string json(const std::vector<PrimerBind>& allp, const std::vector<PcrProduct>& pcr, const std::vector<std::string>& pName,
                   const std::vector<std::string>& pSeq, const std::vector<std::string>& msg, uint32_t distance) const {
    std::string o = "{\"errors\": [";
    bool errors = false;
    for (size_t i = 0; i < msg.size(); ++i) {  // silica.h:106-119
      bool err = msg[i].compare(0, 5, "Error") == 0;
      errors |= err;
      JsonObject e;
      e.str("type", err ? "error" : "warning");
      e.str("title", msg[i]);
      if (i) o.push_back(',');
      o += e.dump();
    }
    o.push_back(']');
    if (!errors) {
      o += ",\"meta\":";
      {  // silica.h:126-134
        JsonObject meta;
        meta.str("version", "0.5.1");
        meta.str("subcommand", "search");
        meta.u64("distance", distance);
        meta.str("genome", c.genome);
        meta.str("outfile", c.outfile);
        meta.u64("maxmatches", c.max_locations);
        meta.boolean("hamming", !c.indel);
        o += meta.dump() + ",";
      }
      o += "\"data\":{\"primers\":[";
      for (size_t i = 0; i < allp.size(); ++i) {  // silica.h:138-153
        const PrimerBind& p = allp[i];
        if (i) o.push_back(',');
        JsonObject j;
        j.str("Chrom", seqname[p.refIndex]);
        j.u64("Id", i);
        j.f64("Tm", p.temp, num(p.temp));
        j.u64("Pos", p.pos + 1);
        j.u64("End", p.pos + pSeq[p.primerId].size());
        j.str("Ori", p.onFor ? "forward" : "reverse");
        j.str("Name", pName[p.primerId]);
        j.f64("MatchTm", p.perfTemp, num(p.perfTemp));
        j.str("Seq", pSeq[p.primerId]);
        j.str("Genome", p.genome);
        o += j.dump();
      }
      o += "],\"amplicons\":[";
      for (size_t i = 0; i < pcr.size(); ++i) {  // silica.h:156-181
        const PcrProduct& a = pcr[i];
        if (i) o.push_back(',');
        // faidx_fetch_seq(chr, forPos, revPos + len - 1): inclusive, clipped to the sequence
        uint64_t cstart = 0;
        for (uint32_t r = 0; r < a.refIndex; ++r) cstart += seqlen[r];
        uint64_t clen = seqlen[a.refIndex] - 1;
        uint64_t b0 = a.forPos, e0 = (uint64_t)a.revPos + pSeq[a.revId].size() - 1;
        std::string seqstr;
        if (b0 < clen) {
          if (e0 >= clen) e0 = clen - 1;
          if (e0 >= b0) seqstr = text->substr(cstart + b0, e0 - b0 + 1);
        }
        JsonObject j;
        j.str("Chrom", seqname[a.refIndex]);
        j.u64("Id", i);
        j.u64("Length", a.leng);
        j.f64("Penalty", a.penalty, num(a.penalty));
        j.u64("ForPos", a.forPos + 1);
        j.u64("ForEnd", a.forPos + pSeq[a.forId].size());
        j.f64("ForTm", a.forTemp, num(a.forTemp));
        j.str("ForName", pName[a.forId]);
        j.str("ForSeq", pSeq[a.forId]);
        j.u64("RevPos", a.revPos + 1);
        j.u64("RevEnd", a.revPos + pSeq[a.revId].size());
        j.f64("RevTm", a.revTemp, num(a.revTemp));
        j.str("RevName", pName[a.revId]);
        j.str("RevSeq", pSeq[a.revId]);
        j.str("Seq", seqstr);
        o += j.dump();
      }
      o += "]}";
    }
    o += "}\n";
    return o;
  }

  // silica.h:355-640; `lines` = the primer FASTA split at '\n'.  Returns the JSON; rc = process exit code.
  std::string run(const std::vector<std::string>& lines, int& rc) {
    std::vector<PrimerBind> allp;
    std::vector<PcrProduct> pcrColl;
    std::vector<std::string> msg, pName, pSeq;
    const Csa& fm_index = *fm;
    const uint32_t nseq = (uint32_t)seqlen.size();
    rc = 0;
    // ---- primer FASTA (silica.h:355-410).  A record is taken only if it is longer than k; the sequence buffer is
    // cleared only when a record is taken (so a too-short record leaks into the next one — kept as is).
    std::string fan, tmpfasta;
    bool fail = false;
    auto take = [&]() -> bool {  // returns false on the fatal error path
      std::string qr = tmpfasta.substr(tmpfasta.size() - c.kmer);
      if (!c.pruneprimer || fm_index.count((const u8*)qr.data(), qr.size()) <= c.maxPruneCount) {
        qr = reverse_complement(qr);
        if (!c.pruneprimer || fm_index.count((const u8*)qr.data(), qr.size()) <= c.maxPruneCount) {
          std::string inseq = replace_non_dna(tmpfasta, msg);
          if (inseq.size() < 10 || inseq.size() < c.kmer) {
            msg.push_back("Error: Input sequence is shorter than 10 nucleotides or shorter than the selected k-mer length!");
            return false;
          }
          if (c.distance >= inseq.size()) {
            c.distance = (uint32_t)inseq.size() - 1;
            msg.push_back("Warning: Distance was adjusted to sequence length!");
          }
          pName.push_back(fan);
          pSeq.push_back(inseq);
        }
      }
      return true;
    };
    for (const std::string& line : lines) {
      if (line.empty()) continue;
      if (line[0] == '>') {
        if (!fan.empty() && !tmpfasta.empty() && tmpfasta.size() > c.kmer) {
          if (!take()) {
            fail = true;
            break;
          }
          tmpfasta = "";
        }
        fan = line.substr(1);
      } else {
        std::string up = line;
        for (auto& ch : up) ch = (char)std::toupper((unsigned char)ch);
        tmpfasta += up;
      }
    }
    if (!fail && !fan.empty() && !tmpfasta.empty() && tmpfasta.size() > c.kmer)
      if (!take()) fail = true;
    if (fail) {
      rc = 1;
      return json(allp, pcrColl, pName, pSeq, msg, c.distance);
    }
    size_t pre_context = c.indel ? c.distance : 0, post_context = pre_context;  // silica.h:417-422
    std::vector<std::vector<PrimerBind>> forBind(nseq), revBind(nseq);
    const Score sc{0, -1, -1, -1};
    for (uint32_t primerId = 0; primerId < pSeq.size(); ++primerId) {
      std::string forQuery = pSeq[primerId], revQuery = reverse_complement(pSeq[primerId]);
      double t0;
      int e1, e2;
      if (!thal(forQuery.c_str(), revQuery.c_str(), &t0, &e1, &e2) || t0 == -999999.0) {  // silica.h:437-442
        msg.push_back("Error: Thermodynamical calculation failed!");
        rc = 1;
        return json(allp, pcrColl, pName, pSeq, msg, c.distance);
      }
      const double matchTemp = t0;
      std::string sequence = pSeq[primerId];
      const uint32_t koffset = (uint32_t)sequence.size() - c.kmer;
      sequence = sequence.substr(sequence.size() - c.kmer);
      std::string revSequence = reverse_complement(sequence);
      std::vector<std::set<std::string>> fwrv(2);
      fwrv[0] = neighbors(sequence, "ACGT", (int)c.distance, c.indel, c.maxNeighborhood);
      fwrv[1] = neighbors(revSequence, "ACGT", (int)c.distance, c.indel, c.maxNeighborhood);
      if (fwrv[0].size() >= c.maxNeighborhood || fwrv[1].size() >= c.maxNeighborhood) {
        std::string x = std::to_string(c.maxNeighborhood);
        msg.push_back("Warning: Neighborhood size exceeds " + x + " candidates. Only first " + x +
                      " neighbors are searched, results are likely incomplete!");
      }
      uint64_t hits = 0;
      for (uint32_t fr = 0; fr < 2; ++fr) {
        std::set<std::pair<uint32_t, uint32_t>> uphit;
        for (auto it = fwrv[fr].begin(); it != fwrv[fr].end() && hits < c.max_locations; ++it) {
          const std::string& query = *it;
          const size_t m = query.size();
          uint64_t occs = fm_index.count((const u8*)query.data(), m);
          if (!occs) continue;
          std::vector<uint64_t> loc = fm_index.locate((const u8*)query.data(), m);
          std::sort(loc.begin(), loc.end());
          for (uint64_t i = 0; i < std::min<uint64_t>(occs, c.max_locations) && hits < c.max_locations; ++i) {
            int64_t best = (int64_t)loc[i], cumsum = 0;
            uint32_t ref = 0;
            for (; ref + 1 < seqlen.size() && best >= cumsum + (int64_t)seqlen[ref]; ++ref) cumsum += seqlen[ref];
            uint32_t chrpos = (uint32_t)(best - cumsum);
            size_t pre_x = pre_context, post_x = post_context;
            if (fr) post_x += koffset;  // silica.h:482-483
            else pre_x += koffset;
            if (pre_x > loc[i]) pre_x = loc[i];
            if (loc[i] + m + post_x > fm_index.n) post_x = fm_index.n - loc[i] - m;
            std::string s = fm_index.extract(loc[i] - pre_x, loc[i] + m + post_x - 1);
            std::string pre = s.substr(0, pre_x);
            s = s.substr(pre_x);
            size_t nl = pre.find_last_of('\n');
            if (nl != std::string::npos) pre = pre.substr(nl + 1);
            std::string post = s.substr(m);
            post = post.substr(0, post.find_first_of('\n'));
            std::string genomicseq = pre + s.substr(0, m) + post;
            if (pre.size() <= chrpos) chrpos -= (uint32_t)pre.size();  // silica.h:501 (non-strict, unlike hunt)
            std::string primer = fr ? forQuery : revQuery;
            std::string searchSeq = fr ? revSequence : sequence;
            double tt;
            if (!thal(primer.c_str(), genomicseq.c_str(), &tt, &e1, &e2) || tt == -999999.0) {  // silica.h:511-516
              msg.push_back("Error: Thermodynamical calculation failed!");
              rc = 1;
              return json(allp, pcrColl, pName, pSeq, msg, c.distance);
            }
            if (tt > c.cutTemp) {
              uint32_t alignpos = chrpos;
              Alignment al;
              needle_free_vertical_ends(genomicseq, searchSeq, al, sc);
              bool lead = true;
              uint32_t stop = (uint32_t)al.row1.size() - trail_gap(al);
              for (uint32_t j = 0; j < stop; ++j) {
                if (al.row1[j] != '-') lead = false;
                if (lead) ++alignpos;
              }
              if (uphit.find(std::make_pair(ref, alignpos)) == uphit.end()) {
                uphit.insert(std::make_pair(ref, alignpos));
                if (fr) {  // silica.h:538-549
                  uint32_t alignshift = alignpos - chrpos;
                  chrpos = alignpos;
                  genomicseq = genomicseq.substr(alignshift, primer.size());
                } else {
                  uint32_t alignshift = alignpos - chrpos;
                  chrpos = alignpos - koffset;
                  if (alignshift >= koffset) {
                    alignshift -= koffset;
                    genomicseq = genomicseq.substr(alignshift, primer.size());
                  }
                }
                PrimerBind prim;
                prim.refIndex = ref;
                prim.temp = tt;
                prim.perfTemp = matchTemp;
                prim.primerId = primerId;
                prim.genome = genomicseq;
                prim.onFor = !fr;
                prim.pos = chrpos;
                (fr ? revBind : forBind)[ref].push_back(prim);
              }
            }
            ++hits;
          }
        }
      }
      if (hits >= c.max_locations) {
        std::string x = std::to_string(c.max_locations);
        msg.push_back("Warning: More than " + x + " matches found. Only first " + x +
                      " matches are reported, results are likely incomplete!");
      }
    }
    for (uint32_t r = 0; r < nseq; ++r) {  // silica.h:581-584
      allp.insert(allp.end(), forBind[r].begin(), forBind[r].end());
      allp.insert(allp.end(), revBind[r].begin(), revBind[r].end());
    }
    std::sort(allp.begin(), allp.end());  // silica.h:587
    if (!c.pruneprimer) {
      for (uint32_t r = 0; r < nseq; ++r) {  // silica.h:592-634
        std::vector<std::pair<uint32_t, uint32_t>> rvByPos;
        for (uint32_t k = 0; k < revBind[r].size(); ++k) rvByPos.push_back(std::make_pair(revBind[r][k].pos, k));
        std::sort(rvByPos.begin(), rvByPos.end());
        std::vector<uint32_t> rvPos(rvByPos.size());
        for (uint32_t k = 0; k < rvByPos.size(); ++k) rvPos[k] = rvByPos[k].first;
        std::vector<uint32_t> cand;
        for (auto fw = forBind[r].begin(); fw != forBind[r].end(); ++fw) {
          auto loIt = std::upper_bound(rvPos.begin(), rvPos.end(), fw->pos);
          auto hiIt = rvPos.end();
          uint64_t hiBound = (uint64_t)fw->pos + (uint64_t)c.maxProdSize;
          if (hiBound < ((uint64_t)1 << 32)) hiIt = std::upper_bound(rvPos.begin(), rvPos.end(), (uint32_t)hiBound);
          cand.clear();
          for (auto pit = loIt; pit != hiIt; ++pit) cand.push_back(rvByPos[pit - rvPos.begin()].second);
          std::sort(cand.begin(), cand.end());
          for (uint32_t ci : cand) {
            const PrimerBind& rv = revBind[r][ci];
            if (rv.pos > fw->pos && rv.pos + pSeq[rv.primerId].size() - fw->pos <= c.maxProdSize) {
              PcrProduct pp;
              pp.refIndex = r;
              pp.forPos = fw->pos;
              pp.forTemp = fw->temp;
              pp.forId = fw->primerId;
              pp.revPos = rv.pos;
              pp.revTemp = rv.temp;
              pp.revId = rv.primerId;
              pp.leng = (uint32_t)((rv.pos + pSeq[pp.revId].size()) - fw->pos);
              double pen = (fw->perfTemp - fw->temp) * c.penDiff;  // silica.h:623-629
              if (pen < 0) pen = 0;
              double bpen = (rv.perfTemp - rv.temp) * c.penDiff;
              if (bpen > 0) pen += bpen;
              pen += std::abs(fw->temp - rv.temp) * c.penMis;
              pen += pp.leng * c.penLen;
              pp.penalty = pen;
              if (c.cutofPen < 0 || pen < c.cutofPen) pcrColl.push_back(pp);
            }
          }
        }
      }
      std::sort(pcrColl.begin(), pcrColl.end());  // silica.h:637
    }
    return json(allp, pcrColl, pName, pSeq, msg, c.distance);
  }
};

}  // namespace orc
