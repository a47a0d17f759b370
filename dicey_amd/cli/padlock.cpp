// `dicey padlock` on the MI355X search path: padlock-probe design over exons (reference src/padlock.h:147-701,
// src/gtf.h:88-274).  The reference walks every exon position and, per position, calls thal() three times, sdsl::count
// four times and count() on two neighbourhoods.  Here the host collects a batch of exons, gets all per-position values
// from ONE library call (dg_padlock_scan: thal of arms and probes, exact and neighbourhood occurrence counts, staged on
// the GPU) and then replays the reference's per-position decision sequence (including the skip after an accepted
// probe) on those arrays, so that the TSV/JSON rows are the reference's, in its order.  GTF parsing, barcode
// assignment and output formatting stay on the host.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <map>
#include <set>

#include "../../include/dicey_gpu.h"
#include "cli_common.hpp"

namespace {

struct PadlockConfig {  // padlock.h:41-78
  bool json = false, indel = true, armMode = true, overlapping = false, computeAll = false, inputFasta = false, absent = false;
  uint32_t distance = 1, armlen = 20, tmdiff = 2;
  double temp = 37.0, mv = 50.0, dv = 1.5, dna_conc = 50.0, dntp = 0.6, mingcth = 0.4, maxgcth = 0.6;
  std::string ucscDB, anchor = "TGCGTCTATTTAGTGGAGCC", spacerleft = "TCCTC", spacerright = "TCTTT", feature = "exon", idname = "gene_id";
  std::set<std::string> geneset;
  std::vector<std::string> chrname;
  std::map<std::string, int32_t> nchr;
  std::string gtfFile, barcodes, primer3Config = "./src/primer3_config/", outfile = "out.tsv", jsonfile, genome, infile;
};

struct GeneInfo {  // gtf.h:21-29
  bool pcoding;
  std::string id, symbol, barcode, code;
  GeneInfo(bool p, const std::string& i, const std::string& s) : pcoding(p), id(i), symbol(s), barcode("NNNNNNNNNNNNNNNNNNNN"), code("000000") {}
};
struct IntervalLabel {  // gtf.h:32-44
  int32_t start, end;
  char strand;
  int32_t lid;
};

std::string now_stamp() {  // boost::posix_time::to_simple_string(second_clock::local_time())
  std::time_t t = std::time(nullptr);
  char b[64];
  std::strftime(b, sizeof b, "%Y-%b-%d %H:%M:%S", std::localtime(&t));
  return b;
}

void guess_ucsc_db(PadlockConfig& c) {  // padlock.h:80-94
  std::string fn = strip_last_extension(c.genome);
  size_t slash = fn.find_last_of('/');
  if (slash != std::string::npos) fn = fn.substr(slash + 1);
  static const char* tab[][2] = {{"GRCh37", "hg19"}, {"GRCh38", "hg38"}, {"GRCz10", "danRer10"}, {"WBcel235", "ce11"}, {"BDGP6", "dm6"},
                                 {"GRCm38", "mm10"}, {"MEDAKA1", "oryLat2"}, {"TAIR10", "hub_1936559_araTha1"},
                                 {"Saccharomyces_cerevisiae.R64", "sacCer3"}};
  c.ucscDB = "Unknown";
  for (auto& e : tab)
    if (fn.find(e[0]) != std::string::npos) {
      c.ucscDB = e[1];
      return;
    }
}

// boost::tokenizer<char_separator<char>>: split at any of the separator characters, empty tokens dropped
std::vector<std::string> split_drop_empty(const std::string& s, const char* seps) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < s.size()) {
    size_t j = s.find_first_of(seps, i);
    if (j == std::string::npos) j = s.size();
    if (j > i) out.push_back(s.substr(i, j - i));
    i = j + 1;
  }
  return out;
}
std::string trimmed(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && std::isspace((unsigned char)s[a])) ++a;
  while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}
std::string strip_quotes(const std::string& v) { return v.size() >= 3 ? v.substr(1, v.size() - 2) : v; }

char complement_iupac(char n) {  // util.h:54-91
  static const char* from = "AaCcGgTtUuRrYySsWwKkMmBbVvDdHhNn";
  static const char* to = "TtGgCcAaAaYyRrSsWwMmKkVvBbHhDdNn";
  const char* p = std::strchr(from, n);
  return (p && n) ? to[p - from] : 'N';
}
void revcomplement(std::string& s) {  // util.h:93-97
  for (char& ch : s) ch = complement_iupac(ch);
  std::reverse(s.begin(), s.end());
}
double gccontent(const char* s, size_t n) {  // util.h:99-107
  if (!n) return -1;
  uint32_t gc = 0;
  for (size_t i = 0; i < n; ++i) {
    const char ch = s[i];
    if (ch == 'N' || ch == 'n') return -1;
    else if (ch == 'C' || ch == 'G' || ch == 'c' || ch == 'g') ++gc;
  }
  return (double)gc / (double)n;
}

// Sequences of a FASTA file as faidx serves them: name = header up to the first whitespace, residues as stored
bool read_fasta(const std::string& path, std::vector<std::string>& names, std::vector<std::string>& seqs) {
  LineReader r(path);
  if (!r.ok()) return false;
  std::string line;
  while (r.next(line)) {
    if (!line.empty() && line[0] == '>') {
      size_t e = line.find_first_of(" \t\r", 1);
      names.push_back(line.substr(1, e == std::string::npos ? std::string::npos : e - 1));
      seqs.emplace_back();
    } else if (!seqs.empty()) {
      if (line.find_first_of("\r \t") == std::string::npos) seqs.back() += line;  // the usual case: residues only
      else
        for (char ch : line)
          if (!(ch == '\r' || ch == ' ' || ch == '\t')) seqs.back().push_back(ch);
    }
  }
  return !names.empty();
}

// gtf.h:88-227.  A malformed line stops the parse with a message; the caller carries on with what was read (parseGTF
// ignores the return value of parseGTFAll).
void parse_gtf_all(const PadlockConfig& c, std::vector<std::vector<IntervalLabel>>& regs, std::vector<GeneInfo>& geneInfo) {
  std::cout << '[' << now_stamp() << "] " << "GTF feature parsing" << std::endl;
  std::map<std::string, int32_t> idMap;
  LineReader r(c.gtfFile);
  std::string gline;
  while (r.ok() && r.next(gline)) {
    if (gline.size() && gline[0] == '#') continue;
    std::vector<std::string> tk = split_drop_empty(gline, "\t");
    if (tk.empty()) continue;
    auto chr = c.nchr.find(tk[0]);
    if (chr == c.nchr.end()) continue;
    if (tk.size() < 3) {
      std::cerr << "Corrupted GTF file!" << std::endl;
      return;
    }
    if (tk[2] != c.feature) continue;
    if (tk.size() == 3) continue;
    if (tk.size() < 7) {  // start, end, score, strand must be there (the reference reads them unchecked up to the strand)
      std::cerr << "Corrupted GTF file!" << std::endl;
      return;
    }
    const int32_t start = (int32_t)std::strtol(tk[3].c_str(), nullptr, 10), end = (int32_t)std::strtol(tk[4].c_str(), nullptr, 10);
    const char strand = tk[6][0];
    if (tk.size() < 9) continue;
    std::vector<std::string> at = split_drop_empty(tk[8], ";");
    std::vector<std::vector<std::string>> kvs;
    for (const std::string& raw : at) kvs.push_back(split_drop_empty(trimmed(raw), " "));
    for (const auto& kv : kvs) {
      if (kv.size() < 2 || kv[0] != c.idname) continue;
      bool includeExon = false;  // exon of a protein-coding transcript (gtf.h:156-169)
      for (const auto& k2 : kvs)
        if (k2.size() >= 2 && k2[0] == "transcript_biotype" && strip_quotes(k2[1]) == "protein_coding") includeExon = true;
      const std::string ensgene = strip_quotes(kv[1]);
      if (!(includeExon && (c.computeAll || c.geneset.count(ensgene)))) continue;
      int32_t idval = (int32_t)geneInfo.size();
      auto it = idMap.find(ensgene);
      if (it == idMap.end()) {
        idMap.insert(std::make_pair(ensgene, idval));
        bool pCode = false;
        std::string symbol = "n.a.";
        for (const auto& k2 : kvs) {
          if (k2.size() < 2) continue;
          if (k2[0] == "gene_biotype" && strip_quotes(k2[1]) == "protein_coding") pCode = true;
          if (k2[0] == "gene_name") symbol = strip_quotes(k2[1]);
        }
        geneInfo.push_back(GeneInfo(pCode, ensgene, symbol));
      } else idval = it->second;
      if (start == 0) {
        std::cerr << "GTF is 1-based format!" << std::endl;
        return;
      }
      if (start > end) {
        std::cerr << "Feature start is greater than feature end!" << std::endl;
        return;
      }
      regs[chr->second].push_back(IntervalLabel{start - 1, end, strand, idval});
    }
  }
}

// gtf.h:230-258: union of a gene's exons per chromosome (touching intervals join), genes in id order
void parse_gtf(const PadlockConfig& c, std::vector<std::vector<IntervalLabel>>& gRegions, std::vector<GeneInfo>& geneInfo) {
  std::vector<std::vector<IntervalLabel>> over(gRegions.size());
  parse_gtf_all(c, over, geneInfo);
  for (size_t ref = 0; ref < over.size(); ++ref) {
    auto& v = over[ref];
    std::stable_sort(v.begin(), v.end(), [](const IntervalLabel& a, const IntervalLabel& b) { return a.lid < b.lid; });
    for (size_t i = 0; i < v.size();) {
      size_t j = i;
      std::vector<std::pair<uint32_t, uint32_t>> iv;
      for (; j < v.size() && v[j].lid == v[i].lid; ++j) iv.emplace_back((uint32_t)v[j].start, (uint32_t)v[j].end);
      std::sort(iv.begin(), iv.end());
      std::vector<std::pair<uint32_t, uint32_t>> merged;
      for (auto& x : iv) {
        if (!merged.empty() && x.first <= merged.back().second) merged.back().second = std::max(merged.back().second, x.second);
        else merged.push_back(x);
      }
      for (auto& x : merged) gRegions[ref].push_back(IntervalLabel{(int32_t)x.first, (int32_t)x.second, v[i].strand, v[i].lid});
      i = j;
    }
  }
}

struct Region {  // one exon interval, strand-corrected
  uint32_t ref;
  IntervalLabel iv;
  std::string seq, rseq;  // exonseq / rexonseq of padlock.h:314-320
  // per arm window p (0 .. len - armlen) / per probe start k: slices of the dg_padlock_result of the batch
  const double *armGC = nullptr, *armTM = nullptr, *probeGC = nullptr, *probeTM = nullptr;
  const int64_t *ucount = nullptr, *nbhits = nullptr;
};

struct Runner {
  PadlockConfig& c;
  dg_index* ix = nullptr;
  dg_thal* th = nullptr;
  std::vector<GeneInfo> geneInfo;
  std::vector<double> barTM;  // per gene
  uint32_t maxNeighborHits = 1, expSeqHits = 1;
  std::ostringstream of, rc;
  bool firstRec = true;
  dg_padlock_result* scan = nullptr;
  explicit Runner(PadlockConfig& cfg) : c(cfg) {}

  bool thal_batch(const std::vector<std::pair<std::string, std::string>>& pairs, std::vector<double>& temps) {
    temps.assign(pairs.size(), 0.0);
    if (pairs.empty()) return true;
    std::string buf;
    std::vector<uint64_t> off(1, 0);
    for (auto& p : pairs) {
      buf += p.first;
      off.push_back(buf.size());
      buf += p.second;
      off.push_back(buf.size());
    }
    if (dg_thal_batch(th, (const uint8_t*)buf.data(), off.data(), pairs.size(), temps.data(), nullptr, nullptr) != DG_OK) {
      std::cerr << "dicey: " << dg_last_error() << std::endl;
      return false;
    }
    return true;
  }

  // One library call for a batch of exons; false = library error (already reported)
  bool compute(std::vector<Region>& regs) {
    std::string buf;
    std::vector<uint64_t> off(1, 0);
    for (auto& R : regs) {
      buf += R.seq;
      off.push_back(buf.size());
    }
    dg_padlock_params pp;
    pp.armlen = c.armlen;
    pp.distance = c.distance;
    pp.hamming = c.indel ? 0 : 1;
    pp.tmdiff = c.tmdiff;
    pp.gc_min = c.mingcth;
    pp.gc_max = c.maxgcth;
    if (scan) dg_padlock_result_free(scan);
    scan = nullptr;
    if (dg_padlock_scan(ix, th, &pp, (const uint8_t*)buf.data(), off.data(), regs.size(), &scan) != DG_OK) {
      std::cerr << "dicey: " << dg_last_error() << std::endl;
      return false;
    }
    for (size_t r = 0; r < regs.size(); ++r) {
      const uint64_t o = scan->pos_off[r];
      regs[r].armGC = scan->arm_gc + o;
      regs[r].armTM = scan->arm_tm + o;
      regs[r].probeGC = scan->probe_gc + o;
      regs[r].probeTM = scan->probe_tm + o;
      regs[r].ucount = scan->arm_count + o;
      regs[r].nbhits = scan->arm_nbcount + o;
    }
    return true;
  }

  static double need(double v) {
    if (v == DG_PADLOCK_NOT_COMPUTED) {
      std::cerr << "dicey: internal error: a probe reached a stage whose values were not computed" << std::endl;
      std::exit(3);
    }
    return v;
  }
  static int64_t need(int64_t v) {
    if (v < 0) {
      std::cerr << "dicey: internal error: a probe reached a stage whose counts were not computed" << std::endl;
      std::exit(3);
    }
    return v;
  }

  // The reference's per-position decisions (padlock.h:321-520) on the precomputed values.  Returns 1 where the reference
  // returns 1 ("Thermodynamical calculation failed"), 0 otherwise.
  int replay(const Region& R) {
    const uint32_t L = c.armlen, T = 2 * c.armlen, exonlen = (uint32_t)R.seq.size();
    const double minGC = c.mingcth, maxGC = c.maxgcth, armTMDiff = c.tmdiff;
    const GeneInfo& gi = geneInfo[R.iv.lid];
    for (uint32_t k = 0; k < (exonlen - T + 1); ++k) {
      const double arm1GC = R.armGC[k];
      if (arm1GC < minGC || arm1GC > maxGC) continue;
      const double arm1TM = need(R.armTM[k]);
      if (arm1TM == -999999.0) return thal_failed();
      double armTMMax = 93 + arm1GC - 675.0 / L;
      if (arm1TM > armTMMax) continue;
      const double arm2GC = R.armGC[k + L];
      if (arm2GC < minGC || arm2GC > maxGC) continue;
      const double arm2TM = need(R.armTM[k + L]);
      if (arm2TM == -999999.0) return thal_failed();
      armTMMax = 93 + arm2GC - 675.0 / L;
      if (arm2TM > armTMMax || std::abs(arm1TM - arm2TM) > armTMDiff) continue;
      const double probeGC = R.probeGC[k];
      if (probeGC < minGC || probeGC > maxGC) continue;
      const double probeTM = need(R.probeTM[k]);
      if (probeTM == -999999.0) return thal_failed();
      const double probeTMMin = 81.5 + probeGC - 675.0 / (2 * L), probeTMMax = probeTMMin + 10;
      if (probeTM < probeTMMin || probeTM > probeTMMax) continue;
      const uint64_t ucount1 = (uint64_t)need(R.ucount[k]);
      if (c.armMode && ucount1 > expSeqHits) continue;
      const uint64_t ucount2 = (uint64_t)need(R.ucount[k + L]);
      if (c.armMode && ucount2 > expSeqHits) continue;
      if (!c.armMode && ucount1 > expSeqHits && ucount2 > expSeqHits) continue;
      if (c.distance > 0) {
        // hits[0] + hits[1] of the reference stop growing once they exceed the threshold; the comparisons see the same
        const bool many1 = (uint64_t)need(R.nbhits[k]) > maxNeighborHits;
        if (c.armMode && many1) continue;
        const bool many2 = (uint64_t)need(R.nbhits[k + L]) > maxNeighborHits;
        if (c.armMode && many2) continue;
        if (!c.armMode && many1 && many2) continue;
      }
      const std::string arm1 = R.seq.substr(k, L), arm2 = R.seq.substr(k + L, L);
      const std::string rarm1 = R.rseq.substr(exonlen - L - k, L), rarm2 = R.rseq.substr(exonlen - L - (k + L), L);
      const std::string padlock = rarm1 + c.spacerleft + c.anchor + gi.barcode + c.spacerright + rarm2;
      const double padlockGC = gccontent(padlock.data(), padlock.size());
      if (padlockGC < minGC || padlockGC > maxGC) continue;
      const double barGC = gccontent(gi.barcode.data(), gi.barcode.size());
      if (barTM[R.iv.lid] == -999999.0) return thal_failed();
      const double bTM = barTM[R.iv.lid];
      int32_t startpos = R.iv.start + (int32_t)k + 1;
      if (R.iv.strand == '-') startpos = R.iv.end - (int32_t)k - (int32_t)T + 1;
      const std::string& chr = c.chrname[R.ref];
      std::ostringstream ucsc;
      if (c.inputFasta) ucsc << "n.a.";
      else ucsc << "https://genome.ucsc.edu/cgi-bin/hgTracks?db=" << c.ucscDB << "&position=" << chr << ":" << startpos << "-" << startpos + T - 1;
      const std::string sl = c.spacerleft.size() ? c.spacerleft : "n.a.", an = c.anchor.size() ? c.anchor : "n.a.";
      const std::string sr = c.spacerright.size() ? c.spacerright : "n.a.";
      of << gi.id << '\t' << gi.symbol << '\t' << gi.code << '\t' << chr << ':' << startpos << '\t' << ucsc.str() << '\t' << R.iv.strand << '\t';
      of << chr << ':' << R.iv.start + 1 << '-' << R.iv.end << '\t' << arm1 << '-' << arm2 << '\t' << sl << '\t' << an << '\t' << gi.barcode << '\t';
      of << sr << '\t' << padlock << '\t' << arm1TM << '\t' << arm2TM << '\t' << bTM << '\t' << probeTM << '\t';
      of << arm1GC << '\t' << arm2GC << '\t' << barGC << '\t' << probeGC << std::endl;
      if (c.json) {
        if (!firstRec) rc << ',';
        else firstRec = false;
        rc << "[\"" << gi.id << "\", \"" << gi.symbol << "\", \"" << gi.code << "\", \"" << chr << ':' << startpos << "\", \"" << ucsc.str() << "\", ";
        rc << "\"" << R.iv.strand << "\", \"" << chr << ':' << R.iv.start + 1 << '-' << R.iv.end << "\", \"" << arm1 << '-' << arm2 << "\", ";
        rc << "\"" << sl << "\", \"" << an << "\", \"" << gi.barcode << "\", \"" << sr << "\", \"" << padlock << "\", ";
        rc << "\"" << arm1TM << "\", \"" << arm2TM << "\", \"" << bTM << "\", \"" << probeTM << "\", ";
        rc << "\"" << arm1GC << "\", \"" << arm2GC << "\", \"" << barGC << "\", \"" << probeGC << "\"]";
      }
      if (!c.overlapping) k += T - 1;
    }
    return 0;
  }
  static int thal_failed() {
    std::cerr << "Error: Thermodynamical calculation failed!" << std::endl;
    return 1;
  }
};

void write_outputs(const PadlockConfig& c, Runner& run) {
  std::ofstream ofile(c.outfile.c_str());
  ofile << run.of.str();
  ofile.close();
  if (c.json) {
    std::ofstream trunc(c.jsonfile.c_str(), std::ios::binary | std::ios::trunc);
    trunc.close();
    append_gzip_member(c.jsonfile, run.rc.str());
  }
}

int run_padlock(PadlockConfig& c) {  // padlock.h:147-531
  Runner run(c);
  run.maxNeighborHits = 1;
  if (c.indel) run.maxNeighborHits = 2 * c.distance;
  if (c.inputFasta && c.absent) run.maxNeighborHits = 0;
  run.expSeqHits = (c.inputFasta && c.absent) ? 0 : 1;
  {  // primer3 config directory (padlock.h:163-176)
    struct stat st;
    if (stat(c.primer3Config.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) {
      std::cerr << "Error: Cannot find primer3 config directory!" << std::endl;
      return 1;
    }
    while (c.primer3Config.size() > 1 && c.primer3Config.back() == '/') c.primer3Config.pop_back();
    c.primer3Config.push_back('/');
    if (stat((c.primer3Config + "tetraloop.dh").c_str(), &st) != 0) {
      std::cerr << "Error: Config directory path appears to be incorrect!" << std::endl;
      return 1;
    }
  }
  // sequences the exons are cut from: the genome, or the input FASTA (padlock.h:310-312)
  std::vector<std::string> srcname, srcseq;
  if (!read_fasta(c.inputFasta ? c.infile : c.genome, srcname, srcseq)) {
    std::cerr << "Error: Cannot read " << (c.inputFasta ? c.infile : c.genome) << std::endl;
    return 1;
  }
  std::vector<std::vector<IntervalLabel>> gRegions(c.nchr.size());
  if (c.inputFasta) {  // gtf.h:260-272
    int32_t runningId = 0;
    for (size_t ref = 0; ref < c.chrname.size(); ++ref) {
      gRegions[ref].push_back(IntervalLabel{0, (int32_t)srcseq[ref].size(), '+', runningId++});
      run.geneInfo.push_back(GeneInfo(true, c.chrname[ref], c.chrname[ref]));
    }
  } else {
    parse_gtf(c, gRegions, run.geneInfo);
    std::set<std::string> gtfSet;
    for (auto& g : run.geneInfo) gtfSet.insert(g.id);
    for (auto& g : c.geneset)
      if (!gtfSet.count(g)) {
        std::cerr << "Error: Gene/transcript name does not exist in GTF file or the transcript biotype is not protein coding: " << g << std::endl;
        return 1;
      }
  }
  // barcodes (padlock.h:215-256)
  std::cout << '[' << now_stamp() << "] " << "Load barcodes" << std::endl;
  uint32_t numBarcodes = 0;
  {
    LineReader r(c.barcodes);
    std::string line, colcode, barcode;
    uint64_t lcount = 0;
    while (r.ok() && r.next(line)) {
      if (line.empty()) continue;
      if (lcount % 2 == 0) {
        if (line[0] == '>') colcode = line.back() == '\r' ? line.substr(1, line.size() - 2) : line.substr(1);
      } else {
        barcode = line.back() == '\r' ? line.substr(0, line.size() - 1) : line;
        for (char& ch : barcode) ch = (char)std::toupper((unsigned char)ch);
        if (numBarcodes < run.geneInfo.size()) {
          run.geneInfo[numBarcodes].barcode = barcode;
          run.geneInfo[numBarcodes].code = colcode;
          ++numBarcodes;
        } else break;
      }
      ++lcount;
    }
  }
  if (numBarcodes < run.geneInfo.size())
    std::cerr << "Warning: only " << numBarcodes << " barcodes available for " << run.geneInfo.size() << " genes!" << std::endl;

  // index + thal on the GPU (padlock.h:177-189, 258-264)
  const int device = std::getenv("DICEY_DEVICE") ? std::atoi(std::getenv("DICEY_DEVICE")) : 0;
  const std::string index_file = strip_last_extension(c.genome) + ".fm9";
  struct stat ist;
  const bool big = stat(index_file.c_str(), &ist) == 0 && ist.st_size > (64 << 20);
  if (dg_index_open(index_file.c_str(), device, ((big || std::getenv("DICEY_KMER_K")) ? (std::getenv("DICEY_FULL_TABLE") ? DG_OPEN_BIG_TABLE : DG_OPEN_DEFAULT) : DG_OPEN_NO_KMER_TABLE) | DG_OPEN_COMPACT, &run.ix) != DG_OK) {
    std::cerr << "Error: FM-Index cannot be loaded!" << std::endl;
    return 1;
  }
  if (dg_thal_open(c.primer3Config.c_str(), c.mv, c.dv, c.dntp, c.dna_conc, device, &run.th) != DG_OK) {
    std::cerr << "dicey: " << dg_last_error() << std::endl;
    dg_index_close(run.ix);
    return 1;
  }
  auto finish = [&](int code) {
    write_outputs(c, run);
    if (run.scan) dg_padlock_result_free(run.scan);
    dg_thal_close(run.th);
    dg_index_close(run.ix);
    return code;
  };
  // JSON header (padlock.h:273-299); nlohmann dump() orders the meta keys alphabetically
  if (c.json) {
    run.rc << "{\"errors\": [],\"meta\":{\"armlength\":" << c.armlen << ",\"barcodes\":" << jstr(c.barcodes) << ",\"distance\":" << c.distance;
    run.rc << ",\"genome\":" << jstr(c.genome) << ",\"gtf\":" << jstr(c.gtfFile) << ",\"hamming\":" << (c.indel ? "false" : "true");
    run.rc << ",\"infile\":" << jstr(c.infile) << ",\"jsonfile\":" << jstr(c.jsonfile) << ",\"outfile\":" << jstr(c.outfile);
    run.rc << ",\"subcommand\":\"padlock\",\"version\":\"" << kVersion << "\"},\"data\":{\"columns\": [";
    run.rc << "\"Gene\", \"Symbol\", \"Code\", \"Position\", \"UCSC\", \"Strand\", \"FeatureCoordinates\", \"ProbeSeq\", \"SpacerLeft\", "
              "\"AnchorSeq\", \"BarcodeSeq\", \"SpacerRight\", \"PadlockSeq\", \"Arm1TM\", \"Arm2TM\", \"BarcodeTM\", \"ProbeTM\", \"Arm1GC\", "
              "\"Arm2GC\", \"BarcodeGC\", \"ProbeGC\"";
    run.rc << "]," << std::endl << "\"rows\": [" << std::endl;
  }
  std::cout << '[' << now_stamp() << "] " << "Compute padlocks" << std::endl;
  run.of << "Gene\tSymbol\tCode\tPosition\tUCSC\tStrand\tFeatureCoordinates\tProbeSeq\tSpacerLeft\tAnchorSeq\tBarcodeSeq\tSpacerRight\tPadlockSeq\t"
            "Arm1TM\tArm2TM\tBarcodeTM\tProbeTM\tArm1GC\tArm2GC\tBarcodeGC\tProbeGC"
         << std::endl;
  // Tm of every gene's barcode against its complement (padlock.h:436-448)
  {
    std::vector<std::pair<std::string, std::string>> pairs;
    for (auto& g : run.geneInfo) {
      std::string rb(g.barcode);
      revcomplement(rb);
      pairs.emplace_back(g.barcode, rb);
    }
    if (!run.thal_batch(pairs, run.barTM)) return finish(2);
  }
  // exons in the reference's order, in batches of about a million positions
  const uint32_t targetlen = 2 * c.armlen;
  std::vector<Region> batch;
  uint64_t batch_pos = 0;
  auto flush = [&]() -> int {
    if (batch.empty()) return 0;
    if (!run.compute(batch)) return 2;
    for (const Region& R : batch) {
      int rcode = run.replay(R);
      if (rcode) return rcode;
    }
    batch.clear();
    batch_pos = 0;
    return 0;
  };
  for (uint32_t refIndex = 0; refIndex < c.nchr.size(); ++refIndex) {
    // chromosome name -> sequence of the source FASTA (faidx_fetch_seq by name)
    int64_t src = -1;
    for (size_t i = 0; i < srcname.size(); ++i)
      if (srcname[i] == c.chrname[refIndex]) {
        src = (int64_t)i;
        break;
      }
    for (size_t i = 0; i < gRegions[refIndex].size(); ++i) {
      const IntervalLabel& iv = gRegions[refIndex][i];
      if (src < 0) {
        std::cerr << "Warning: Could not fetch sequence for " << c.chrname[refIndex] << ":" << iv.start + 1 << "-" << iv.end << "!" << std::endl;
        continue;
      }
      const std::string& s = srcseq[(size_t)src];
      int64_t b = iv.start, e = iv.end;  // faidx_fetch_seq clips to the sequence
      if (b < 0) b = 0;
      if (e > (int64_t)s.size()) e = (int64_t)s.size();
      Region R;
      R.ref = refIndex;
      R.iv = iv;
      if (b < e) R.seq = s.substr((size_t)b, (size_t)(e - b));
      for (char& ch : R.seq) ch = (char)std::toupper((unsigned char)ch);
      if (iv.strand == '-') revcomplement(R.seq);
      if (R.seq.size() < targetlen) continue;
      R.rseq = R.seq;
      revcomplement(R.rseq);
      batch_pos += R.seq.size();
      batch.push_back(std::move(R));
      if (batch_pos >= (1u << 20)) {
        int rcode = flush();
        if (rcode) return finish(rcode);
      }
    }
  }
  {
    int rcode = flush();
    if (rcode) return finish(rcode);
  }
  if (c.json) run.rc << "]}}";
  finish(0);
  std::cout << '[' << now_stamp() << "] Done." << std::endl;
  return 0;
}

const OptSpec kPadlockOpts[] = {
    {"help", '?', false},      {"genome", 'g', true},      {"gtf", 't', true},         {"config", 'i', true},     {"outfile", 'o', true},
    {"json", 'j', true},       {"absent", 'e', false},     {"hamming", 'n', false},    {"anchor", 'a', true},     {"spacerleft", 'l', true},
    {"spacerright", 'r', true}, {"barcodes", 'b', true},   {"distance", 'd', true},    {"armlen", 'm', true},     {"tmdiff", 'z', true},
    {"gcmin", 0, true},        {"gcmax", 0, true},         {"attribute", 'u', true},   {"feature", 'f', true},    {"probe", 'p', false},
    {"overlapping", 'v', false}, {"enttemp", 0, true},     {"monovalent", 0, true},    {"divalent", 0, true},     {"dna", 0, true},
    {"dntp", 0, true},         {"infile", 0, true}};

void padlock_usage(const char* sub) {  // padlock.h:595-602
  std::cout << "Usage:" << std::endl;
  std::cout << "Probes for one gene: dicey " << sub << " [OPTIONS] -g <ref.fa.gz> -t <ref.gtf.gz> -b <barcodes.fa.gz> ENSG00000171862" << std::endl;
  std::cout << "Probes for one transcript: dicey " << sub << " [OPTIONS] -u transcript_id -g <ref.fa.gz> -t <ref.gtf.gz> -b <barcodes.fa.gz> ENST00000406757" << std::endl;
  std::cout << "Probes for a set of genes: dicey " << sub << " [OPTIONS] -g <ref.fa.gz> -t <ref.gtf.gz> -b <barcodes.fa.gz> <gene.list.file>" << std::endl;
  std::cout << "Probes for custom FASTA input: dicey " << sub << " [OPTIONS] -g <ref.fa.gz> -t <ref.gtf.gz> -b <barcodes.fa.gz> <sequences.fa>" << std::endl;
  std::cout << "\nGeneric options:\n"
               "  -? [ --help ]                          show help message\n"
               "  -g [ --genome ] arg                    genome file\n"
               "  -t [ --gtf ] arg                       gtf/gff3 file\n"
               "  -i [ --config ] arg (=./src/primer3_config/) primer3 config directory\n"
               "  -o [ --outfile ] arg (=out.tsv)        output file\n"
               "  -j [ --json ] arg                      gzipped JSON file [optional]\n"
               "  -e [ --absent ]                        source sequence is absent in reference genome [FASTA input]\n"
               "  -n [ --hamming ]                       use hamming neighborhood instead of edit distance\n"
               "\nPadlock options:\n"
               "  -a [ --anchor ] arg (=TGCGTCTATTTAGTGGAGCC) anchor sequence\n"
               "  -l [ --spacerleft ] arg (=TCCTC)       spacer left\n"
               "  -r [ --spacerright ] arg (=TCTTT)      spacer right\n"
               "  -b [ --barcodes ] arg                  FASTA barcode file\n"
               "  -d [ --distance ] arg (=1)             neighborhood distance\n"
               "  -m [ --armlen ] arg (=20)              probe arm length\n"
               "  -z [ --tmdiff ] arg (=2)               Tm difference between arms\n"
               "  --gcmin arg (=0.4)                     minimum arm GC fraction\n"
               "  --gcmax arg (=0.6)                     maximum arm GC fraction\n"
               "  -u [ --attribute ] arg (=gene_id)      gtf/gff3 attribute\n"
               "  -f [ --feature ] arg (=exon)           gtf/gff3 feature\n"
               "  -p [ --probe ]                         apply distance to entire probe, i.e., only one arm needs to be unique\n"
               "  -v [ --overlapping ]                   allow overlapping probes\n"
               "\nParameters for Tm Calculation:\n"
               "  --enttemp arg (=37)                    temperature for entropie and entalpie calculation in Celsius\n"
               "  --monovalent arg (=50)                 concentration of monovalent ions in mMol\n"
               "  --divalent arg (=1.5)                  concentration of divalent ions in mMol\n"
               "  --dna arg (=50)                        concentration of annealing(!) Oligos in nMol\n"
               "  --dntp arg (=0.6)                      the sum  of all dNTPs in mMol\n\n";
}

}  // namespace

int padlock_main(int argc, char** argv) {  // padlock.h:533-698
  PadlockConfig c;
  Parsed p = parse_options(argc, argv, kPadlockOpts, sizeof kPadlockOpts / sizeof kPadlockOpts[0]);
  if (!p.error.empty()) {
    std::cerr << "dicey: " << p.error << std::endl;
    return 2;
  }
  bool help = false, has_genome = false, has_bar = false, has_gtf = false, has_infile = false;
  for (auto& kv : p.kv) {
    const std::string& k = kv.first;
    const std::string& v = kv.second;
    if (k == "help") help = true;
    else if (k == "genome") c.genome = v, has_genome = true;
    else if (k == "gtf") c.gtfFile = v, has_gtf = true;
    else if (k == "config") c.primer3Config = v;
    else if (k == "outfile") c.outfile = v;
    else if (k == "json") c.jsonfile = v, c.json = true;
    else if (k == "absent") c.absent = true;
    else if (k == "hamming") c.indel = false;
    else if (k == "anchor") c.anchor = v;
    else if (k == "spacerleft") c.spacerleft = v;
    else if (k == "spacerright") c.spacerright = v;
    else if (k == "barcodes") c.barcodes = v, has_bar = true;
    else if (k == "distance") c.distance = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
    else if (k == "armlen") c.armlen = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
    else if (k == "tmdiff") c.tmdiff = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
    else if (k == "gcmin") c.mingcth = std::strtod(v.c_str(), nullptr);
    else if (k == "gcmax") c.maxgcth = std::strtod(v.c_str(), nullptr);
    else if (k == "attribute") c.idname = v;
    else if (k == "feature") c.feature = v;
    else if (k == "probe") c.armMode = false;
    else if (k == "overlapping") c.overlapping = true;
    else if (k == "enttemp") c.temp = std::strtod(v.c_str(), nullptr);  // only enters thal()'s dG output, which temponly skips
    else if (k == "monovalent") c.mv = std::strtod(v.c_str(), nullptr);
    else if (k == "divalent") c.dv = std::strtod(v.c_str(), nullptr);
    else if (k == "dna") c.dna_conc = std::strtod(v.c_str(), nullptr);
    else if (k == "dntp") c.dntp = std::strtod(v.c_str(), nullptr);
    else if (k == "infile") c.infile = v, has_infile = true;
  }
  if (!p.positional.empty()) {
    c.infile = p.positional.back();
    has_infile = true;
  }
  if (help || !has_infile || !has_genome || !has_bar || !has_gtf) {
    padlock_usage(argv[0]);
    return -1;
  }
  if (!file_nonempty(c.genome)) {
    std::cerr << "Error: Genome does not exist!" << std::endl;
    return 1;
  }
  guess_ucsc_db(c);
  if (!file_nonempty(c.gtfFile)) {
    std::cerr << "Error: GTF file does not exist!" << std::endl;
    return 1;
  }
  if (!file_nonempty(c.barcodes)) {
    std::cerr << "Error: Barcode FASTA file does not exist!" << std::endl;
    return 1;
  }
  if (!file_nonempty(c.infile)) {  // padlock.h:631-657
    if (c.infile == "all") c.computeAll = true;
    else c.geneset.insert(c.infile);
  } else if (is_fasta(c.infile)) c.inputFasta = true;
  else {
    std::ifstream geneFile(c.infile.c_str());
    std::string gline;
    while (geneFile.good()) {
      std::getline(geneFile, gline);
      std::vector<std::string> tk = split_drop_empty(gline, " \t,;");
      if (!tk.empty()) c.geneset.insert(tk[0]);
    }
  }
  {  // chromosome names in faidx order (padlock.h:659-683)
    std::vector<uint32_t> lens;
    std::vector<std::string> names;
    if (!seq_len_name(c.inputFasta ? c.infile : c.genome, lens, names)) {
      std::cerr << "Error: Cannot read the sequence names of " << (c.inputFasta ? c.infile : c.genome) << std::endl;
      return 1;
    }
    c.chrname = names;
    for (size_t i = 0; i < names.size(); ++i) c.nchr.insert(std::make_pair(names[i], (int32_t)i));
  }
  std::cout << '[' << now_stamp() << "] dicey ";
  for (int i = 0; i < argc; ++i) std::cout << argv[i] << ' ';
  std::cout << std::endl;
  return run_padlock(c);
}
