// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under dicey_amd/ may include, link or call this.
//
// CPU restatement of `dicey padlock` (reference src/padlock.h:147-531 runPadlock, src/gtf.h:88-274 GTF handling,
// src/util.h:93-107 revcomplement / gccontent).  primer3thal::thal() is NOT restated: it is the reference's own
// thal.h compiled in place (oracle/_ref) and passed in as a function pointer, already initialised with the run's
// thermodynamic parameters (padlock.h:177-189).  Doubles are written with operator<< of a default std::ostream, as the
// reference does.  File access (faidx, gzip) is replaced by strings handed in by the caller.  Parity of this driver
// logic is unpinned: the reference has no tests and its binary cannot be built here.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "fm9.hpp"
#include "hunt_ref.hpp"
#include "search_ref.hpp"

namespace orc {

struct PadlockParams {  // padlock.h:41-78 with the defaults of :537-575
  bool json = false, indel = true, armMode = true, overlapping = false, computeAll = false, inputFasta = false, absent = false;
  uint32_t distance = 1, armlen = 20, tmdiff = 2;
  double mingcth = 0.4, maxgcth = 0.6;
  std::string ucscDB = "Unknown", anchor = "TGCGTCTATTTAGTGGAGCC", spacerleft = "TCCTC", spacerright = "TCTTT";
  std::string feature = "exon", idname = "gene_id";
  std::set<std::string> geneset;
  std::string genome, infile, outfile = "out.tsv", barcodes, gtf, jsonfile;  // only echoed into the JSON meta block
};

struct GeneInfo {  // gtf.h:21-29
  bool pcoding;
  std::string id, symbol, barcode = "NNNNNNNNNNNNNNNNNNNN", code = "000000";
};
struct IntervalLabel {  // gtf.h:32-44
  int32_t start, end;
  char strand;
  int32_t lid;
};

// boost::tokenizer<char_separator<char>>(s, sep): split at any separator character, empty tokens dropped
inline std::vector<std::string> tokens_of(const std::string& s, const char* seps) {
  std::vector<std::string> out;
  std::string cur;
  for (char ch : s) {
    if (std::strchr(seps, ch)) {
      if (!cur.empty()) out.push_back(cur);
      cur.clear();
    } else cur.push_back(ch);
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}
inline std::string trim_copy(const std::string& s) {  // boost::trim: std::isspace at both ends
  size_t a = 0, b = s.size();
  while (a < b && std::isspace((unsigned char)s[a])) ++a;
  while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}
inline std::string unquote(const std::string& v) { return v.size() >= 3 ? v.substr(1, v.size() - 2) : v; }  // gtf.h:170

inline void revcomplement_iupac(std::string& s) {  // util.h:93-97
  for (char& ch : s) ch = complement_base(ch);
  std::reverse(s.begin(), s.end());
}
inline double gccontent(const std::string& s) {  // util.h:99-107
  if (s.empty()) return -1;
  uint32_t gc = 0;
  for (char ch : s) {
    if (ch == 'N' || ch == 'n') return -1;
    else if (ch == 'C' || ch == 'G' || ch == 'c' || ch == 'g') ++gc;
  }
  return (double)gc / (double)s.size();
}

struct PadlockRun {
  const Csa* fm;
  ThalFn thal;
  PadlockParams c;
  std::vector<std::string> chrname;        // faidx order of the genome (or of the input FASTA, padlock.h:656-680)
  std::vector<std::string> chrseq;         // what faidx_fetch_seq reads from: the sequences as stored in the FASTA
  std::string err;                         // what the reference prints to std::cerr

  // gtf.h:88-227.  Returns false where the reference prints a message and stops parsing (the caller carries on with what
  // was read so far, exactly like parseGTF ignores parseGTFAll's return value).
  bool parse_gtf_all(const std::vector<std::string>& lines, std::vector<std::vector<IntervalLabel>>& regs, std::vector<GeneInfo>& geneInfo) {
    std::map<std::string, int32_t> nchr;
    for (size_t i = 0; i < chrname.size(); ++i) nchr.insert(std::make_pair(chrname[i], (int32_t)i));
    std::map<std::string, int32_t> idMap;
    for (const std::string& gline : lines) {
      if (gline.size() && gline[0] == '#') continue;
      std::vector<std::string> tk = tokens_of(gline, "\t");
      size_t t = 0;
      if (tk.empty()) continue;
      const std::string chrName = tk[t++];
      if (nchr.find(chrName) == nchr.end()) continue;
      const int32_t chrid = nchr[chrName];
      if (t == tk.size()) { err += "Corrupted GTF file!\n"; return false; }
      ++t;
      if (t >= tk.size()) { err += "Corrupted GTF file!\n"; return false; }
      const std::string ft = tk[t++];
      if (ft != c.feature) continue;
      if (t == tk.size()) continue;
      if (t + 1 >= tk.size()) { err += "Corrupted GTF file!\n"; return false; }  // (the reference would read past the end here)
      const int32_t start = (int32_t)std::stol(tk[t++]);
      const int32_t end = (int32_t)std::stol(tk[t++]);
      ++t;  // score
      if (t >= tk.size()) { err += "Corrupted GTF file!\n"; return false; }
      const char strand = tk[t++][0];
      ++t;  // frame
      if (t >= tk.size()) continue;  // (undefined in the reference)
      const std::string attr = tk[t];
      std::vector<std::string> at = tokens_of(attr, ";");
      for (const std::string& raw : at) {
        std::vector<std::string> kv = tokens_of(trim_copy(raw), " ");
        if (kv.empty() || kv[0] != c.idname || kv.size() < 2) continue;
        bool includeExon = false;  // protein-coding transcript? (gtf.h:156-169)
        for (const std::string& r2 : at) {
          std::vector<std::string> k2 = tokens_of(trim_copy(r2), " ");
          if (k2.size() >= 2 && k2[0] == "transcript_biotype" && unquote(k2[1]) == "protein_coding") includeExon = true;
        }
        const std::string ensgene = unquote(kv[1]);
        if (!(includeExon && (c.computeAll || c.geneset.count(ensgene)))) continue;
        int32_t idval = (int32_t)geneInfo.size();
        auto it = idMap.find(ensgene);
        if (it == idMap.end()) {
          idMap.insert(std::make_pair(ensgene, idval));
          GeneInfo gi;
          gi.pcoding = false;
          gi.id = ensgene;
          gi.symbol = "n.a.";
          for (const std::string& r2 : at) {
            std::vector<std::string> k2 = tokens_of(trim_copy(r2), " ");
            if (k2.size() < 2) continue;
            if (k2[0] == "gene_biotype" && unquote(k2[1]) == "protein_coding") gi.pcoding = true;
            if (k2[0] == "gene_name") gi.symbol = unquote(k2[1]);
          }
          geneInfo.push_back(gi);
        } else idval = it->second;
        if (start == 0) { err += "GTF is 1-based format!\n"; return false; }
        if (start > end) { err += "Feature start is greater than feature end!\n"; return false; }
        regs[chrid].push_back(IntervalLabel{start - 1, end, strand, idval});
      }
    }
    return true;
  }

  // gtf.h:230-258: per chromosome and gene the union of its exons (boost::icl::interval_set joins overlapping and touching
  // right-open intervals); genes in id order, intervals ascending; strand = strand of the gene's first record
  void parse_gtf(const std::vector<std::string>& lines, std::vector<std::vector<IntervalLabel>>& gRegions, std::vector<GeneInfo>& geneInfo) {
    std::vector<std::vector<IntervalLabel>> over(gRegions.size());
    parse_gtf_all(lines, over, geneInfo);
    for (size_t ref = 0; ref < over.size(); ++ref) {
      std::stable_sort(over[ref].begin(), over[ref].end(), [](const IntervalLabel& a, const IntervalLabel& b) { return a.lid < b.lid; });
      size_t i = 0;
      while (i < over[ref].size()) {
        size_t j = i;
        std::vector<std::pair<uint32_t, uint32_t>> iv;
        while (j < over[ref].size() && over[ref][j].lid == over[ref][i].lid) {
          iv.emplace_back((uint32_t)over[ref][j].start, (uint32_t)over[ref][j].end);
          ++j;
        }
        std::sort(iv.begin(), iv.end());
        std::vector<std::pair<uint32_t, uint32_t>> merged;
        for (auto& x : iv) {
          if (!merged.empty() && x.first <= merged.back().second) merged.back().second = std::max(merged.back().second, x.second);
          else merged.push_back(x);
        }
        for (auto& x : merged) gRegions[ref].push_back(IntervalLabel{(int32_t)x.first, (int32_t)x.second, over[ref][i].strand, over[ref][i].lid});
        i = j;
      }
    }
  }

  // htslib faidx_fetch_seq(fai, name, beg, end_inclusive): clipped to the sequence
  std::string fetch(uint32_t ref, int32_t beg, int32_t end_incl) const {
    const std::string& s = chrseq[ref];
    int64_t b = beg, e = (int64_t)end_incl + 1;
    if (b < 0) b = 0;
    if (e > (int64_t)s.size()) e = (int64_t)s.size();
    if (b >= e) return std::string();
    return s.substr((size_t)b, (size_t)(e - b));
  }

  u64 count(const std::string& s) const { return fm->count((const u8*)s.data(), s.size()); }

  // padlock.h:147-531.  gtf_lines: the (decompressed) GTF; bar_lines: the (decompressed) barcode FASTA.
  int run(const std::vector<std::string>& gtf_lines, const std::vector<std::string>& bar_lines, std::string& tsv, std::string& json) {
    uint32_t maxNeighborHits = 1;
    if (c.indel) maxNeighborHits = 2 * c.distance;
    if (c.inputFasta && c.absent) maxNeighborHits = 0;
    uint32_t expSeqHits = 1;
    if (c.inputFasta && c.absent) expSeqHits = 0;
    const double armTMDiff = c.tmdiff, minGC = c.mingcth, maxGC = c.maxgcth;

    std::vector<std::vector<IntervalLabel>> gRegions(chrname.size());
    std::vector<GeneInfo> geneInfo;
    if (c.inputFasta) {  // gtf.h:260-272
      int32_t runningId = 0;
      for (size_t ref = 0; ref < chrname.size(); ++ref) {
        gRegions[ref].push_back(IntervalLabel{0, (int32_t)chrseq[ref].size(), '+', runningId++});
        GeneInfo gi;
        gi.pcoding = true;
        gi.id = gi.symbol = chrname[ref];
        geneInfo.push_back(gi);
      }
    } else {
      parse_gtf(gtf_lines, gRegions, geneInfo);
      std::set<std::string> gtfSet;
      for (auto& g : geneInfo) gtfSet.insert(g.id);
      for (auto& g : c.geneset)
        if (!gtfSet.count(g)) {
          err += "Error: Gene/transcript name does not exist in GTF file or the transcript biotype is not protein coding: " + g + "\n";
          return 1;
        }
    }
    // barcodes (padlock.h:216-256)
    uint32_t numBarcodes = 0;
    {
      uint64_t lcount = 0;
      std::string colcode, barcode;
      for (const std::string& line : bar_lines) {
        if (line.empty()) continue;
        if (lcount % 2 == 0) {
          if (line[0] == '>') colcode = line.back() == '\r' ? line.substr(1, line.size() - 2) : line.substr(1);
        } else {
          barcode = line.back() == '\r' ? line.substr(0, line.size() - 1) : line;
          for (char& ch : barcode) ch = (char)std::toupper((unsigned char)ch);
          if (numBarcodes < geneInfo.size()) {
            geneInfo[numBarcodes].barcode = barcode;
            geneInfo[numBarcodes].code = colcode;
            ++numBarcodes;
          } else break;
        }
        ++lcount;
      }
    }
    if (numBarcodes < geneInfo.size())
      err += "Warning: only " + std::to_string(numBarcodes) + " barcodes available for " + std::to_string(geneInfo.size()) + " genes!\n";

    std::ostringstream rc, of;
    bool firstRec = true;
    if (c.json) {  // padlock.h:273-299; nlohmann dump(): keys in alphabetical order
      rc << "{\"errors\": [],\"meta\":";
      {  // padlock.h:282-295, through the reference's nlohmann when oracle/_ref is loaded (hunt_ref.hpp JsonObject)
        JsonObject meta;
        meta.str("version", "0.5.1");
        meta.str("subcommand", "padlock");
        meta.u64("armlength", c.armlen);
        meta.u64("distance", c.distance);
        meta.str("genome", c.genome);
        meta.str("infile", c.infile);
        meta.str("outfile", c.outfile);
        meta.str("barcodes", c.barcodes);
        meta.str("gtf", c.gtf);
        meta.str("jsonfile", c.jsonfile);
        meta.boolean("hamming", !c.indel);
        rc << meta.dump() << ",";
      }
      rc << "\"data\":{\"columns\": [";
      rc << "\"Gene\", \"Symbol\", \"Code\", \"Position\", \"UCSC\", \"Strand\", \"FeatureCoordinates\", \"ProbeSeq\", \"SpacerLeft\", "
            "\"AnchorSeq\", \"BarcodeSeq\", \"SpacerRight\", \"PadlockSeq\", \"Arm1TM\", \"Arm2TM\", \"BarcodeTM\", \"ProbeTM\", \"Arm1GC\", "
            "\"Arm2GC\", \"BarcodeGC\", \"ProbeGC\"";
      rc << "]," << std::endl << "\"rows\": [" << std::endl;
    }
    of << "Gene\tSymbol\tCode\tPosition\tUCSC\tStrand\tFeatureCoordinates\tProbeSeq\tSpacerLeft\tAnchorSeq\tBarcodeSeq\tSpacerRight\tPadlockSeq\t"
          "Arm1TM\tArm2TM\tBarcodeTM\tProbeTM\tArm1GC\tArm2GC\tBarcodeGC\tProbeGC"
       << std::endl;
    auto finish = [&](int code) {
      tsv = of.str();
      json = rc.str();
      return code;
    };
    auto tm = [&](const std::string& a, const std::string& b, double& out) {
      double t;
      int e1, e2;
      int ok = thal(a.c_str(), b.c_str(), &t, &e1, &e2);
      out = t;
      return ok && t != -999999.0;
    };
    const uint32_t targetlen = 2 * c.armlen;
    for (uint32_t refIndex = 0; refIndex < chrname.size(); ++refIndex) {
      for (size_t i = 0; i < gRegions[refIndex].size(); ++i) {
        const IntervalLabel& reg = gRegions[refIndex][i];
        std::string exonseq = fetch(refIndex, reg.start, reg.end - 1);
        for (char& ch : exonseq) ch = (char)std::toupper((unsigned char)ch);
        if (reg.strand == '-') revcomplement_iupac(exonseq);
        const uint32_t exonlen = (uint32_t)exonseq.size();
        if (exonlen < targetlen) continue;
        std::string rexonseq(exonseq);
        revcomplement_iupac(rexonseq);
        for (uint32_t k = 0; k < (exonlen - targetlen + 1); ++k) {
          std::string arm1 = exonseq.substr(k, c.armlen);
          double arm1GC = gccontent(arm1);
          if (arm1GC < minGC || arm1GC > maxGC) continue;
          std::string rarm1 = rexonseq.substr(exonlen - c.armlen - k, c.armlen);
          double arm1TM;
          if (!tm(arm1, rarm1, arm1TM)) { err += "Error: Thermodynamical calculation failed!\n"; return finish(1); }
          double armTMMax = 93 + arm1GC - 675.0 / c.armlen;
          if (arm1TM > armTMMax) continue;
          std::string arm2 = exonseq.substr(k + c.armlen, c.armlen);
          double arm2GC = gccontent(arm2);
          if (arm2GC < minGC || arm2GC > maxGC) continue;
          std::string rarm2 = rexonseq.substr(exonlen - c.armlen - (k + c.armlen), c.armlen);
          double arm2TM;
          if (!tm(arm2, rarm2, arm2TM)) { err += "Error: Thermodynamical calculation failed!\n"; return finish(1); }
          armTMMax = 93 + arm2GC - 675.0 / c.armlen;
          if (arm2TM > armTMMax || std::abs(arm1TM - arm2TM) > armTMDiff) continue;
          std::string probe = exonseq.substr(k, targetlen);
          double probeGC = gccontent(probe);
          if (probeGC < minGC || probeGC > maxGC) continue;
          std::string rprobe = rexonseq.substr(exonlen - targetlen - k, targetlen);
          double probeTM;
          if (!tm(probe, rprobe, probeTM)) { err += "Error: Thermodynamical calculation failed!\n"; return finish(1); }
          double probeTMMin = 81.5 + probeGC - 675.0 / (2 * c.armlen);
          double probeTMMax = probeTMMin + 10;
          if (probeTM < probeTMMin || probeTM > probeTMMax) continue;
          // uniqueness (padlock.h:380-389)
          u64 ucount1 = count(arm1) + count(rarm1);
          if (c.armMode && ucount1 > expSeqHits) continue;
          u64 ucount2 = count(arm2) + count(rarm2);
          if (c.armMode && ucount2 > expSeqHits) continue;
          if (!c.armMode && ucount1 > expSeqHits && ucount2 > expSeqHits) continue;
          if (c.distance > 0) {  // padlock.h:392-428
            uint32_t hits[2] = {0, 0}, hitsOther[2] = {0, 0};
            std::set<std::string> fwrv[2];
            fwrv[0] = neighbors(arm1, "ACGT", (int)c.distance, c.indel, 10000);
            fwrv[1] = neighbors(rarm1, "ACGT", (int)c.distance, c.indel, 10000);
            for (int f = 0; f < 2; ++f)
              for (auto it = fwrv[f].begin(); it != fwrv[f].end() && hits[0] + hits[1] <= maxNeighborHits; ++it) hits[f] += (uint32_t)count(*it);
            if (c.armMode && hits[0] + hits[1] > maxNeighborHits) continue;
            fwrv[0] = neighbors(arm2, "ACGT", (int)c.distance, c.indel, 10000);
            fwrv[1] = neighbors(rarm2, "ACGT", (int)c.distance, c.indel, 10000);
            for (int f = 0; f < 2; ++f)
              for (auto it = fwrv[f].begin(); it != fwrv[f].end() && hitsOther[0] + hitsOther[1] <= maxNeighborHits; ++it)
                hitsOther[f] += (uint32_t)count(*it);
            if (c.armMode && hitsOther[0] + hitsOther[1] > maxNeighborHits) continue;
            if (!c.armMode && hits[0] + hits[1] > maxNeighborHits && hitsOther[0] + hitsOther[1] > maxNeighborHits) continue;
          }
          const GeneInfo& gi = geneInfo[reg.lid];
          std::string padlock = rarm1 + c.spacerleft + c.anchor + gi.barcode + c.spacerright + rarm2;
          double padlockGC = gccontent(padlock);
          if (padlockGC < minGC || padlockGC > maxGC) continue;
          std::string bartmp = gi.barcode;
          double barGC = gccontent(bartmp);
          std::string rbartmp(bartmp);
          revcomplement_iupac(rbartmp);
          double barTM;
          if (!tm(bartmp, rbartmp, barTM)) { err += "Error: Thermodynamical calculation failed!\n"; return finish(1); }
          int32_t startpos = reg.start + (int32_t)k + 1;
          if (reg.strand == '-') startpos = reg.end - (int32_t)k - (int32_t)targetlen + 1;
          std::ostringstream ucsc;
          if (c.inputFasta) ucsc << "n.a.";
          else
            ucsc << "https://genome.ucsc.edu/cgi-bin/hgTracks?db=" << c.ucscDB << "&position=" << chrname[refIndex] << ":" << startpos << "-"
                 << startpos + targetlen - 1;
          of << gi.id << '\t' << gi.symbol << '\t' << gi.code << '\t' << chrname[refIndex] << ':' << startpos << '\t' << ucsc.str() << '\t';
          of << reg.strand << '\t' << chrname[refIndex] << ':' << reg.start + 1 << '-' << reg.end << '\t' << arm1 << '-' << arm2 << '\t';
          of << (c.spacerleft.size() ? c.spacerleft : std::string("n.a.")) << '\t' << (c.anchor.size() ? c.anchor : std::string("n.a.")) << '\t';
          of << gi.barcode << '\t' << (c.spacerright.size() ? c.spacerright : std::string("n.a.")) << '\t' << padlock << '\t';
          of << arm1TM << '\t' << arm2TM << '\t' << barTM << '\t' << probeTM << '\t';
          of << arm1GC << '\t' << arm2GC << '\t' << barGC << '\t' << probeGC << std::endl;
          if (c.json) {
            if (!firstRec) rc << ',';
            else firstRec = false;
            rc << "[\"" << gi.id << "\", \"" << gi.symbol << "\", \"" << gi.code << "\", \"" << chrname[refIndex] << ':' << startpos << "\", ";
            rc << "\"" << ucsc.str() << "\", \"" << reg.strand << "\", \"" << chrname[refIndex] << ':' << reg.start + 1 << '-' << reg.end << "\", ";
            rc << "\"" << arm1 << '-' << arm2 << "\", \"" << (c.spacerleft.size() ? c.spacerleft : std::string("n.a.")) << "\", ";
            rc << "\"" << (c.anchor.size() ? c.anchor : std::string("n.a.")) << "\", \"" << gi.barcode << "\", ";
            rc << "\"" << (c.spacerright.size() ? c.spacerright : std::string("n.a.")) << "\", \"" << padlock << "\", ";
            rc << "\"" << arm1TM << "\", \"" << arm2TM << "\", \"" << barTM << "\", \"" << probeTM << "\", ";
            rc << "\"" << arm1GC << "\", \"" << arm2GC << "\", \"" << barGC << "\", \"" << probeGC << "\"]";
          }
          if (!c.overlapping) k += targetlen - 1;
        }
      }
    }
    if (c.json) rc << "]}}";
    return finish(0);
  }
};

}  // namespace orc
