// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under dicey_amd/ may include, link or call this.
//
// CPU restatement of the `dicey hunt` hot path.  PARITY STATUS: the reference has no tests and
// cannot be built in this environment (needs Boost, htslib, sdsl-lite), so this restatement is
// pinned only by the known answers the survey measured from the unmodified reference headers
// (neighbourhood sizes of TCTCTGCACACACGTTGTAC, one needle alignment) — see
// tests/test_oracle_known_answers.py.  Everything else is "parity unpinned".
//
// Each function names the reference lines it follows.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <set>
#include <string>
#include <unordered_set>
#include <vector>

#include "fm9.hpp"

namespace orc {

// ---------------------------------------------------------------------------------------------
// util.h
// ---------------------------------------------------------------------------------------------
// util.h:54-91 complement(): IUPAC table, unknown -> 'N'
inline char complement_base(char b) {
  static const char* from = "AaCcGgTtUuRrYySsWwKkMmBbVvDdHhNn";
  static const char* to = "TtGgCcAaAaYyRrSsWwMmKkVvBbHhDdNn";
  for (int i = 0; from[i]; ++i)
    if (from[i] == b) return to[i];
  return 'N';
}
// util.h:110-114 reverseComplement(): upper-case, complement each, reverse
inline std::string reverse_complement(std::string s) {
  for (auto& ch : s) ch = complement_base((char)std::toupper((unsigned char)ch));
  return std::string(s.rbegin(), s.rend());
}
// util.h:208-219 replaceNonDna(): one warning per replaced character
inline std::string replace_non_dna(const std::string& s, std::vector<std::string>& msg) {
  std::string out;
  for (char ch : s) {
    if (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') out.push_back(ch);
    else {
      msg.push_back("Warning: Non-DNA character in nucleotide sequence detected and replaced by 'N'!");
      out.push_back('N');
    }
  }
  return out;
}

// ---------------------------------------------------------------------------------------------
// neighbors.h:29-92
// ---------------------------------------------------------------------------------------------
struct Neighborhood {
  std::set<std::string> S;
  std::string alphabet;  // iterated in std::set<char> order => sorted
  int input_dist = 0;
  bool indel = false;
  uint32_t maxsize = 0;

  // neighbors.h:29-45 — in indel mode keep only substring-minimal strings
  void put(const std::string& s) {
    if (!indel) {
      S.insert(s);
      return;
    }
    bool keep = true;
    for (auto it = S.begin(); it != S.end();) {
      if (it->find(s) != std::string::npos) it = S.erase(it);
      else {
        if (s.find(*it) != std::string::npos) keep = false;
        ++it;
      }
    }
    if (keep) S.insert(s);
  }
  // neighbors.h:47-83 — order of exploration: delete, keep, substitute (alphabet order),
  // insert-before (alphabet order); bail out whenever the working set has reached maxsize.
  void walk(std::string& q, int dist, int pos) {
    if (S.size() >= maxsize) return;
    if (pos >= (int)q.size()) {
      if (dist < input_dist) put(q);
      return;
    }
    if (dist > 0 && indel) {
      std::string del = q.substr(0, pos) + q.substr(pos + 1);
      walk(del, dist - 1, pos);
    }
    walk(q, dist, pos + 1);
    if (dist > 0) {
      const char orig = q[pos];
      for (char a : alphabet)
        if (a != orig) {
          q[pos] = a;
          walk(q, dist - 1, pos + 1);
        }
      q[pos] = orig;
      if (indel)
        for (char a : alphabet) {
          std::string ins = q.substr(0, pos) + std::string(1, a) + q.substr(pos);
          walk(ins, dist - 1, pos + 1);
        }
    }
  }
};
// neighbors.h:86-92
inline std::set<std::string> neighbors(const std::string& query, const std::string& alphabet, int dist, bool indel,
                                       uint32_t maxsize) {
  Neighborhood nb;
  std::set<char> a(alphabet.begin(), alphabet.end());
  nb.alphabet.assign(a.begin(), a.end());
  nb.input_dist = dist;
  nb.indel = indel;
  nb.maxsize = maxsize;
  std::string q(query);
  nb.put(q);
  nb.walk(q, dist, 0);
  return nb.S;
}

// The same set without the reference's quadratic _insert scan, for parity runs over many distance-2 queries (the literal
// form above needs ~1.3 s per 20-mer).  Only used where the result provably does not depend on the generation order: the
// whole <= d-edit language is enumerated into a hash set and, if it stays below maxsize (so neighbors.h:50 can never fire:
// the working set is a subset of the strings generated so far), the answer is its substring-minimal subset (edit mode) /
// the set itself (Hamming mode).  Otherwise the literal restatement runs.  tests/test_oracle.py holds the two against
// each other.
inline void language_walk(std::string& q, const std::string& alphabet, int input_dist, int dist, bool indel, int pos,
                          std::unordered_set<std::string>& out) {
  if (pos >= (int)q.size()) {
    if (dist < input_dist) out.insert(q);
    return;
  }
  if (dist > 0 && indel) {
    std::string del = q.substr(0, pos) + q.substr(pos + 1);
    language_walk(del, alphabet, input_dist, dist - 1, indel, pos, out);
  }
  language_walk(q, alphabet, input_dist, dist, indel, pos + 1, out);
  if (dist > 0) {
    const char orig = q[pos];
    for (char a : alphabet)
      if (a != orig) {
        q[pos] = a;
        language_walk(q, alphabet, input_dist, dist - 1, indel, pos + 1, out);
      }
    q[pos] = orig;
    if (indel)
      for (char a : alphabet) {
        std::string ins = q.substr(0, pos) + std::string(1, a) + q.substr(pos);
        language_walk(ins, alphabet, input_dist, dist - 1, indel, pos + 1, out);
      }
  }
}
inline std::set<std::string> neighbors_fast(const std::string& query, const std::string& alphabet, int dist, bool indel,
                                            uint32_t maxsize) {
  std::set<char> a(alphabet.begin(), alphabet.end());
  std::string alpha(a.begin(), a.end());
  std::unordered_set<std::string> lang;
  lang.insert(query);
  std::string q(query);
  language_walk(q, alpha, dist, dist, indel, 0, lang);
  if (lang.size() >= maxsize) return neighbors(query, alphabet, dist, indel, maxsize);
  std::set<std::string> out;
  if (!indel) {
    out.insert(lang.begin(), lang.end());
    return out;
  }
  const size_t minlen = query.size() > (size_t)dist ? query.size() - dist : 1;
  for (const std::string& s : lang) {
    bool minimal = true;
    for (size_t len = minlen; len < s.size() && minimal; ++len)
      for (size_t at = 0; at + len <= s.size(); ++at)
        if (lang.count(s.substr(at, len))) {
          minimal = false;
          break;
        }
    if (minimal) out.insert(s);
  }
  return out;
}
inline bool& fast_neighbors_enabled() {
  static bool on = false;
  return on;
}
inline std::set<std::string> neighbors_for_driver(const std::string& query, const std::string& alphabet, int dist, bool indel,
                                                   uint32_t maxsize) {
  return fast_neighbors_enabled() ? neighbors_fast(query, alphabet, dist, indel, maxsize) : neighbors(query, alphabet, dist, indel, maxsize);
}

// ---------------------------------------------------------------------------------------------
// needle.h:59-138 with AlignConfig<false,true> (align.h:43-80) and DnaScore (align.h:11-33)
// rows = a1 (genomic window), cols = a2 (query); vertical moves (gap in the query row) are
// free in column 0 and column n; horizontal moves always cost ge.
// ---------------------------------------------------------------------------------------------
struct Score {
  int match, mismatch, go, ge;
};
struct Alignment {
  std::string row0, row1;  // align[0][*] (a1 with gaps), align[1][*] (a2 with gaps)
};
inline int needle_free_vertical_ends(const std::string& a1, const std::string& a2, Alignment& al, const Score& sc) {
  const size_t m = a1.size(), n = a2.size(), mf = n + 1;
  std::vector<int> s(n + 1, 0);
  std::vector<uint8_t> took_h((m + 1) * (n + 1), 0), took_v((m + 1) * (n + 1), 0);
  auto vgap = [&](size_t col, int cost) { return (col == 0 || col == n) ? 0 : cost; };  // align.h:59-64
  int prevsub = 0;
  for (size_t row = 0; row <= m; ++row)
    for (size_t col = 0; col <= n; ++col) {
      if (row == 0 && col == 0) {
        s[0] = 0;
        prevsub = 0;
      } else if (row == 0) {
        s[col] = (int)col * sc.ge;  // needle.h:86
        took_h[col] = 1;
      } else if (col == 0) {
        s[0] = vgap(0, (int)row * sc.ge);
        prevsub = (row == 1) ? 0 : vgap(0, (int)(row - 1) * sc.ge);
        took_v[row * mf] = 1;
      } else {
        int diag_from = prevsub;
        prevsub = s[col];
        int d = diag_from + (a1[row - 1] == a2[col - 1] ? sc.match : sc.mismatch);
        int v = prevsub + vgap(col, sc.ge);
        int h = s[col - 1] + sc.ge;
        s[col] = std::max(std::max(d, v), h);  // needle.h:105
        if (s[col] == h) took_h[row * mf + col] = 1;  // needle.h:108 (horizontal wins ties)
        else if (s[col] == v) took_v[row * mf + col] = 1;
      }
    }
  std::string r0, r1;
  size_t row = m, col = n;
  while (row > 0 || col > 0) {  // needle.h:118-131
    if (took_h[row * mf + col]) {
      --col;
      r0.push_back('-');
      r1.push_back(a2[col]);
    } else if (took_v[row * mf + col]) {
      --row;
      r0.push_back(a1[row]);
      r1.push_back('-');
    } else {
      --row;
      --col;
      r0.push_back(a1[row]);
      r1.push_back(a2[col]);
    }
  }
  al.row0.assign(r0.rbegin(), r0.rend());  // align.h:176-203
  al.row1.assign(r1.rbegin(), r1.rend());
  return s[n];
}
// hunter.h:69-77
inline uint32_t trail_gap(const Alignment& al) {
  uint32_t len = (uint32_t)al.row1.size(), last = len - 1;
  for (uint32_t j = 0; j < len; ++j)
    if (al.row1[j] != '-') last = j;
  return len - last - 1;
}
// hunter.h:79-88
inline int hamming_score(const std::string& a, const std::string& b, const Score& sc) {
  int score = 0;
  for (size_t i = 0; i < a.size() && i < b.size(); ++i) score += (a[i] == b[i]) ? sc.match : sc.mismatch;
  return score;
}

// ---------------------------------------------------------------------------------------------
// nlohmann::json 3.5.0 dump() of flat objects: keys sorted (std::map), no whitespace,
// strings escaped as serializer::dump_escaped with ensure_ascii=false.
// ---------------------------------------------------------------------------------------------
inline std::string json_str(const std::string& s) {
  static const char* hex = "0123456789abcdef";
  std::string o = "\"";
  for (unsigned char ch : s) {
    switch (ch) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\b': o += "\\b"; break;
      case '\f': o += "\\f"; break;
      case '\n': o += "\\n"; break;
      case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break;
      default:
        if (ch < 0x20) {
          o += "\\u00";
          o.push_back(hex[ch >> 4]);
          o.push_back(hex[ch & 15]);
        } else o.push_back((char)ch);
    }
  }
  o.push_back('"');
  return o;
}

// ---------------------------------------------------------------------------------------------
// The reference serialises every JSON object with nlohmann::json::dump() (hunter.h:112-116,122-152, silica.h:113-181).
// When oracle/_ref/libjsonref.so (that very header, compiled in place) has been loaded with orc_use_ref_json, the writers
// below build their objects THROUGH it — member by member, in the reference's assignment order and with its member types
// — and only the fixed punctuation between objects is written here.  Without it they fall back to json_str / std::to_string
// (same output on everything tests/test_oracle.py::test_own_json_writer_equals_reference_nlohmann tries).
struct RefJsonApi {
  void* (*nw)() = nullptr;
  void (*del)(void*) = nullptr;
  void (*set_str)(void*, const char*, const char*, uint64_t) = nullptr;
  void (*set_u64)(void*, const char*, uint64_t) = nullptr;
  void (*set_i64)(void*, const char*, int64_t) = nullptr;
  void (*set_bool)(void*, const char*, int) = nullptr;
  void (*set_f64)(void*, const char*, double) = nullptr;
  char* (*dump)(void*, uint64_t*) = nullptr;
  void (*release)(char*) = nullptr;
  bool ok() const { return nw && del && set_str && set_u64 && set_i64 && set_bool && set_f64 && dump && release; }
};
inline RefJsonApi& ref_json_api() {
  static RefJsonApi api;
  return api;
}
inline bool& ref_json_enabled() {
  static bool on = false;
  return on;
}
// one JSON object; members in any order (nlohmann keeps a std::map, the fallback sorts the same way)
class JsonObject {
 public:
  JsonObject() {
    if (ref_json_enabled()) h_ = ref_json_api().nw();
  }
  ~JsonObject() {
    if (h_) ref_json_api().del(h_);
  }
  JsonObject(const JsonObject&) = delete;
  void str(const char* k, const std::string& v) {
    if (h_) ref_json_api().set_str(h_, k, v.data(), v.size());
    else own_.emplace_back(k, json_str(v));
  }
  void u64(const char* k, uint64_t v) {
    if (h_) ref_json_api().set_u64(h_, k, v);
    else own_.emplace_back(k, std::to_string(v));
  }
  void i64(const char* k, int64_t v) {
    if (h_) ref_json_api().set_i64(h_, k, v);
    else own_.emplace_back(k, std::to_string(v));
  }
  void boolean(const char* k, bool v) {
    if (h_) ref_json_api().set_bool(h_, k, v);
    else own_.emplace_back(k, v ? "true" : "false");
  }
  // doubles: `formatted` is what the caller's number printer gives (the fallback has no Grisu2 of its own)
  void f64(const char* k, double v, const std::string& formatted) {
    if (h_) ref_json_api().set_f64(h_, k, v);
    else own_.emplace_back(k, formatted);
  }
  std::string dump() {
    if (h_) {
      uint64_t n = 0;
      char* p = ref_json_api().dump(h_, &n);
      if (!p) return "<nlohmann::json::dump() threw: the reference would terminate here>";
      std::string s(p, n);
      ref_json_api().release(p);
      return s;
    }
    std::sort(own_.begin(), own_.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    std::string o = "{";
    for (size_t i = 0; i < own_.size(); ++i) {
      if (i) o.push_back(',');
      o += json_str(own_[i].first) + ":" + own_[i].second;
    }
    return o + "}";
  }

 private:
  void* h_ = nullptr;
  std::vector<std::pair<std::string, std::string>> own_;
};

// ---------------------------------------------------------------------------------------------
// hunter.h:53-66 DnaHit, :99-160 writer, :289-444 per-query loop
// ---------------------------------------------------------------------------------------------
struct DnaHit {
  int32_t score;
  uint32_t chr, start;
  char strand;
  std::string refalign, queryalign;
  bool operator<(const DnaHit& b) const {  // hunter.h:63-65
    return (score > b.score) || (score == b.score && chr < b.chr) || (score == b.score && chr == b.chr && start < b.start);
  }
};
struct HuntParams {
  uint32_t distance = 1;
  bool indel = true;    // !--hamming
  bool reverse = true;  // !--forward
  uint64_t max_locations = 1000;
  uint32_t max_neighborhood = 10000;
  std::string genome, outfile;
};
struct OpCounters {  // SURVEY §8(d) op counts of the reference algorithm
  uint64_t patterns = 0, bs_steps = 0, located = 0, extracted = 0, needles = 0;
};

inline std::string hunt_json(const HuntParams& p, uint32_t distance, const std::string& sequence, const std::string& qname,
                             const std::vector<std::string>& seqname, const std::vector<DnaHit>& ht,
                             const std::vector<std::string>& msg) {
  std::string o = "{\"errors\": [";
  bool errors = false;
  for (size_t i = 0; i < msg.size(); ++i) {  // hunter.h:105-118
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
    {  // hunter.h:122-134
      JsonObject meta;
      meta.str("version", "0.5.1");
      meta.str("subcommand", "hunt");
      meta.u64("distance", distance);
      meta.str("sequence", sequence);
      if (!qname.empty()) meta.str("name", qname);
      meta.str("genome", p.genome);
      meta.str("outfile", p.outfile);
      meta.u64("maxmatches", p.max_locations);
      meta.boolean("hamming", !p.indel);
      meta.boolean("forwardonly", !p.reverse);
      o += meta.dump() + ",";
    }
    o += "\"data\":[";
    uint32_t oldchr = 999999, oldstart = 0;
    bool first = true;
    for (const auto& h : ht) {  // hunter.h:140-155
      if (oldchr != h.chr || oldstart != h.start) {
        if (!first) o.push_back(',');
        first = false;
        uint32_t nuc = 0;
        for (char ch : h.refalign) nuc += (ch != '-');
        JsonObject j;
        j.i64("distance", std::abs(h.score));
        j.str("chr", seqname[h.chr]);
        j.u64("start", h.start);
        j.u64("end", h.start + nuc - 1);
        j.str("strand", std::string(1, h.strand));
        j.str("refalign", h.refalign);
        j.str("queryalign", h.queryalign);
        o += j.dump();
      }
      oldchr = h.chr;
      oldstart = h.start;
    }
    o.push_back(']');
  }
  o += "}\n";
  return o;
}

// One query through hunter.h:291-444.  seqlen[i] = faidx length + 1 (util.h:201).
// `pushed` (optional) receives the hit vector in reference push order, before the sort.
inline std::string hunt_one(const Csa& fm, const std::vector<uint32_t>& seqlen, const std::vector<std::string>& seqname,
                            const HuntParams& p, const std::string& qname_in, const std::string& seq_in,
                            std::vector<DnaHit>* pushed = nullptr, OpCounters* oc = nullptr,
                            std::vector<DnaHit>* sorted = nullptr) {
  std::vector<DnaHit> ht;
  std::vector<std::string> msg;
  std::string sequence = seq_in;
  uint32_t distance = p.distance;
  if (sequence.size() < 10) {
    msg.push_back("Error: Input sequence is shorter than 10 nucleotides!");
    return hunt_json(p, distance, sequence, qname_in, seqname, ht, msg);
  }
  for (auto& ch : sequence) ch = (char)std::toupper((unsigned char)ch);
  sequence = replace_non_dna(sequence, msg);
  std::string rev = reverse_complement(sequence);
  if (distance >= sequence.size()) {
    distance = (uint32_t)sequence.size() - 1;
    msg.push_back("Warning: Distance was adjusted to sequence length!");
  }
  size_t pre_context = p.indel ? distance : 0, post_context = pre_context;
  std::vector<std::set<std::string>> fwrv(2);
  fwrv[0] = neighbors_for_driver(sequence, "ACGT", (int)distance, p.indel, p.max_neighborhood);
  if (p.reverse) fwrv[1] = neighbors_for_driver(rev, "ACGT", (int)distance, p.indel, p.max_neighborhood);
  if (fwrv[0].size() >= p.max_neighborhood || fwrv[1].size() >= p.max_neighborhood) {
    std::string x = std::to_string(p.max_neighborhood);
    msg.push_back("Warning: Neighborhood size exceeds " + x + " candidates. Only first " + x +
                  " neighbors are searched, results are likely incomplete!");
  }
  const Score sc{0, -1, -1, -1};
  uint64_t hits = 0;
  for (uint32_t fr = 0; fr < 2; ++fr) {
    for (auto it = fwrv[fr].begin(); it != fwrv[fr].end() && hits < p.max_locations; ++it) {
      const std::string& query = *it;
      const size_t m = query.size();
      uint64_t occs = fm.count((const u8*)query.data(), m);
      if (oc) {
        ++oc->patterns;
        oc->bs_steps += m;
      }
      if (!occs) continue;
      std::vector<uint64_t> loc = fm.locate((const u8*)query.data(), m);
      std::sort(loc.begin(), loc.end());
      if (oc) oc->located += loc.size();
      for (uint64_t i = 0; i < std::min<uint64_t>(occs, p.max_locations) && hits < p.max_locations; ++i) {
        int64_t best = (int64_t)loc[i], cumsum = 0;
        uint32_t ref = 0;
        for (; ref + 1 < seqlen.size() && best >= cumsum + (int64_t)seqlen[ref]; ++ref) cumsum += seqlen[ref];
        uint32_t chrpos = (uint32_t)(best - cumsum);
        size_t pre_x = pre_context, post_x = post_context;
        if (pre_x > loc[i]) pre_x = loc[i];
        if (loc[i] + m + post_x > fm.n) post_x = fm.n - loc[i] - m;
        std::string s = fm.extract(loc[i] - pre_x, loc[i] + m + post_x - 1);
        if (oc) ++oc->extracted;
        std::string pre = s.substr(0, pre_x);
        s = s.substr(pre_x);
        size_t nl = pre.find_last_of('\n');
        if (nl != std::string::npos) pre = pre.substr(nl + 1);
        std::string post = s.substr(m);
        post = post.substr(0, post.find_first_of('\n'));
        std::string genomic = pre + s.substr(0, m) + post;
        if (pre.size() < chrpos) chrpos -= (uint32_t)pre.size();  // hunter.h:382 (strict)
        const std::string& qq = fr == 0 ? sequence : rev;
        const char strand = fr == 0 ? '+' : '-';
        if (p.indel) {
          Alignment al;
          int score = needle_free_vertical_ends(genomic, qq, al, sc);
          if (oc) ++oc->needles;
          std::string ra, qa;
          bool lead = true;
          uint32_t stop = (uint32_t)al.row1.size() - trail_gap(al);
          for (uint32_t j = 0; j < stop; ++j) {
            if (al.row1[j] != '-') lead = false;
            if (!lead) {
              ra.push_back(al.row0[j]);
              qa.push_back(al.row1[j]);
            } else ++chrpos;
          }
          ht.push_back(DnaHit{score, ref, chrpos + 1, strand, ra, qa});
        } else {
          ht.push_back(DnaHit{hamming_score(genomic, qq, sc), ref, chrpos + 1, strand, genomic, qq});
        }
        ++hits;
      }
    }
  }
  if (hits >= p.max_locations) {
    std::string x = std::to_string(p.max_locations);
    msg.push_back("Warning: More than " + x + " matches found. Only first " + x +
                  " matches are reported, results are likely incomplete!");
  }
  if (pushed) *pushed = ht;
  std::sort(ht.begin(), ht.end());
  if (sorted) *sorted = ht;
  return hunt_json(p, distance, sequence, qname_in, seqname, ht, msg);
}

}  // namespace orc
