// ORACLE — TEST INFRASTRUCTURE ONLY (ctypes entry points for tests/, smoke() and bench.py's
// cpu_baseline leg).  Nothing under dicey_amd/ may include, link or call this.
#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "fm9.hpp"
#include "hunt_ref.hpp"
#include "search_ref.hpp"
#include "padlock_ref.hpp"

using namespace orc;

namespace {
thread_local std::string g_err;
char* dup_out(const std::string& s, uint64_t* len) {
  char* p = (char*)std::malloc(s.size() + 1);
  std::memcpy(p, s.data(), s.size());
  p[s.size()] = 0;
  if (len) *len = s.size();
  return p;
}
struct Handle {
  Csa csa;
};
}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }
void orc_free(void* p) { std::free(p); }

// index.h:97-123 — build a csa_wt<> over `text` (no '\0' inside) and store it in sdsl layout
int orc_build_fm9(const char* text, uint64_t len, const char* out_path) {
  try {
    std::string T(text, len);
    Csa c = build_csa(T);
    write_file(out_path, serialize_csa(c));
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

void* orc_open(const char* path) {
  try {
    std::vector<u8> b = read_file(path);
    Handle* h = new Handle;
    h->csa = parse_csa(b.data(), b.size());
    return h;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void orc_close(void* h) { delete (Handle*)h; }
uint64_t orc_size(void* h) { return ((Handle*)h)->csa.n; }

uint64_t orc_count(void* h, const char* pat, uint64_t m) { return ((Handle*)h)->csa.count((const u8*)pat, m); }
// returns number of occurrences; writes up to cap positions (unsorted, SA order)
uint64_t orc_locate(void* h, const char* pat, uint64_t m, uint64_t* out, uint64_t cap) {
  std::vector<u64> v = ((Handle*)h)->csa.locate((const u8*)pat, m);
  for (uint64_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
  return v.size();
}
void orc_extract(void* h, uint64_t b, uint64_t e, char* out) {
  std::string s = ((Handle*)h)->csa.extract(b, e);
  std::memcpy(out, s.data(), s.size());
}
uint64_t orc_sa(void* h, uint64_t i) { return ((Handle*)h)->csa.sa(i); }
uint32_t orc_code_len(void* h, int ch) { return (uint32_t)(((Handle*)h)->csa.path[(u8)ch] >> 56); }

// brute-force twin: the semantics sdsl::count/locate must have
uint64_t orc_bf_locate(const char* text, uint64_t n, const char* pat, uint64_t m, uint64_t* out, uint64_t cap) {
  uint64_t k = 0;
  if (m == 0 || m > n) return 0;
  for (uint64_t i = 0; i + m <= n; ++i)
    if (std::memcmp(text + i, pat, m) == 0) {
      if (k < cap && out) out[k] = i;
      ++k;
    }
  return k;
}

// Route every JSON object of the hunt / search writers through the reference's own nlohmann::json (oracle/_ref/libjsonref.so,
// built in place from /root/reference/src/jlib by oracle/Makefile).  path == NULL switches back to the restated writer.
// Returns 1 when the reference library is in use.
int orc_use_ref_json(const char* path) {
  ref_json_enabled() = false;
  if (!path) return 0;
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    g_err = dlerror();
    return 0;
  }
  RefJsonApi& a = ref_json_api();
  a.nw = (void* (*)())dlsym(h, "ref_json_new");
  a.del = (void (*)(void*))dlsym(h, "ref_json_free");
  a.set_str = (void (*)(void*, const char*, const char*, uint64_t))dlsym(h, "ref_json_set_str");
  a.set_u64 = (void (*)(void*, const char*, uint64_t))dlsym(h, "ref_json_set_u64");
  a.set_i64 = (void (*)(void*, const char*, int64_t))dlsym(h, "ref_json_set_i64");
  a.set_bool = (void (*)(void*, const char*, int))dlsym(h, "ref_json_set_bool");
  a.set_f64 = (void (*)(void*, const char*, double))dlsym(h, "ref_json_set_f64");
  a.dump = (char* (*)(void*, uint64_t*))dlsym(h, "ref_json_dump");
  a.release = (void (*)(char*))dlsym(h, "ref_json_release");
  if (!a.ok()) {
    g_err = "libjsonref.so lacks the object builder";
    return 0;
  }
  ref_json_enabled() = true;
  return 1;
}
int orc_ref_json_in_use() { return ref_json_enabled() ? 1 : 0; }

// hunt_one's neighbourhoods through neighbors_fast (hash-set minimality, tested equal to the literal restatement)
void orc_fast_neighbors(int on) { fast_neighbors_enabled() = on != 0; }

// neighbors.h:86 — newline-joined, std::set order; fast != 0: the hash-set form
char* orc_neighbors2(const char* query, int dist, int indel, uint32_t maxsize, int fast, uint64_t* count) {
  std::set<std::string> s = fast ? neighbors_fast(query, "ACGT", dist, indel != 0, maxsize) : neighbors(query, "ACGT", dist, indel != 0, maxsize);
  std::string o;
  for (const auto& x : s) {
    o += x;
    o.push_back('\n');
  }
  if (count) *count = s.size();
  return dup_out(o, nullptr);
}
char* orc_neighbors(const char* query, int dist, int indel, uint32_t maxsize, uint64_t* count) {
  std::set<std::string> s = neighbors(query, "ACGT", dist, indel != 0, maxsize);
  std::string o;
  for (const auto& x : s) {
    o += x;
    o.push_back('\n');
  }
  if (count) *count = s.size();
  return dup_out(o, nullptr);
}

// needle.h:59 with AlignConfig<false,true>, DnaScore(0,-1,-1,-1) (hunter.h:383-389)
int orc_needle(const char* a1, const char* a2, char** row0, char** row1, uint32_t* trailgap) {
  Alignment al;
  int sc = needle_free_vertical_ends(a1, a2, al, Score{0, -1, -1, -1});
  *row0 = dup_out(al.row0, nullptr);
  *row1 = dup_out(al.row1, nullptr);
  if (trailgap) *trailgap = trail_gap(al);
  return sc;
}

struct orc_hunt_params {
  uint32_t distance;
  int32_t hamming;
  int32_t forward_only;
  uint64_t max_locations;
  uint32_t max_neighborhood;
};

static HuntParams mk(const orc_hunt_params* p, const char* genome, const char* outfile) {
  HuntParams hp;
  hp.distance = p->distance;
  hp.indel = !p->hamming;
  hp.reverse = !p->forward_only;
  hp.max_locations = p->max_locations;
  hp.max_neighborhood = p->max_neighborhood;
  hp.genome = genome ? genome : "";
  hp.outfile = outfile ? outfile : "";
  return hp;
}

// hunter.h:291-444 over a batch; returns the concatenated JSON lines.
// If hits_blob != NULL it also returns, per query, the hit vector in PUSH order (pre-sort) as text
// lines "qi\tscore\tchr\tstart\tstrand\trefalign\tqueryalign\n".
char* orc_hunt(void* h, const uint32_t* seqlen, const char* const* seqname, uint32_t nseq, const orc_hunt_params* p,
               const char* genome, const char* outfile, const char* const* qnames, const char* const* seqs, uint64_t nq,
               uint64_t* json_len, char** hits_blob, uint64_t* hits_len) {
  const Csa& fm = ((Handle*)h)->csa;
  std::vector<uint32_t> sl(seqlen, seqlen + nseq);
  std::vector<std::string> sn(nseq);
  for (uint32_t i = 0; i < nseq; ++i) sn[i] = seqname[i];
  HuntParams hp = mk(p, genome, outfile);
  std::string json, blob;
  for (uint64_t qi = 0; qi < nq; ++qi) {
    std::vector<DnaHit> pushed;
    json += hunt_one(fm, sl, sn, hp, qnames ? qnames[qi] : "", seqs[qi], hits_blob ? &pushed : nullptr);
    if (hits_blob)
      for (const auto& d : pushed)
        blob += std::to_string(qi) + "\t" + std::to_string(d.score) + "\t" + std::to_string(d.chr) + "\t" +
                std::to_string(d.start) + "\t" + d.strand + "\t" + d.refalign + "\t" + d.queryalign + "\n";
  }
  if (hits_blob) *hits_blob = dup_out(blob, hits_len);
  return dup_out(json, json_len);
}

// CPU baseline: time the restated reference path on `threads` host threads (the reference itself is
// single-threaded).  Returns wall seconds; counters summed over threads.
double orc_hunt_timed(void* h, const uint32_t* seqlen, uint32_t nseq, const orc_hunt_params* p, const char* const* seqs,
                      uint64_t nq, uint32_t threads, uint64_t* counters /*[5]*/, uint64_t* total_hits) {
  const Csa& fm = ((Handle*)h)->csa;
  std::vector<uint32_t> sl(seqlen, seqlen + nseq);
  std::vector<std::string> sn(nseq, "chr");
  HuntParams hp = mk(p, "", "");
  if (threads == 0) threads = 1;
  std::vector<OpCounters> ocs(threads);
  std::vector<uint64_t> th(threads, 0);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (uint32_t t = 0; t < threads; ++t)
    pool.emplace_back([&, t]() {
      for (uint64_t qi = t; qi < nq; qi += threads) {
        std::vector<DnaHit> pushed;
        std::string js = hunt_one(fm, sl, sn, hp, "", seqs[qi], &pushed, &ocs[t]);
        th[t] += pushed.size() + (js.size() == 0);
      }
    });
  for (auto& th_ : pool) th_.join();
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (counters) {
    for (int i = 0; i < 5; ++i) counters[i] = 0;
    for (auto& o : ocs) {
      counters[0] += o.patterns;
      counters[1] += o.bs_steps;
      counters[2] += o.located;
      counters[3] += o.extracted;
      counters[4] += o.needles;
    }
  }
  if (total_hits) {
    *total_hits = 0;
    for (auto v : th) *total_hits += v;
  }
  return dt;
}

struct orc_search_params {
  int32_t hamming, pruneprimer;
  double cutTemp;
  uint32_t maxProdSize;
  double cutofPen, penDiff, penMis, penLen;
  uint32_t kmer, distance, maxNeighborhood, maxPruneCount;
  uint64_t max_locations;
};

// silica.h:355-640 + writer :100-187.  thal_fn / dump_fn come from oracle/_ref (the reference's own thal.h / json.hpp).
char* orc_search(void* h, const uint32_t* seqlen, const char* const* seqname, uint32_t nseq, const char* text, uint64_t textlen,
                 const orc_search_params* p, const char* genome, const char* outfile, const char* fasta, void* thal_fn, void* dump_fn,
                 int* rc, uint64_t* json_len) {
  SearchRun r;
  r.fm = &((Handle*)h)->csa;
  r.seqlen.assign(seqlen, seqlen + nseq);
  for (uint32_t i = 0; i < nseq; ++i) r.seqname.push_back(seqname[i]);
  std::string T(text, textlen);
  r.text = &T;
  r.thal = (ThalFn)thal_fn;
  r.dump_double = (DumpDoubleFn)dump_fn;
  r.c.indel = !p->hamming;
  r.c.pruneprimer = p->pruneprimer != 0;
  r.c.cutTemp = p->cutTemp;
  r.c.maxProdSize = p->maxProdSize;
  r.c.cutofPen = p->cutofPen;
  r.c.penDiff = p->penDiff;
  r.c.penMis = p->penMis;
  r.c.penLen = p->penLen;
  r.c.kmer = p->kmer;
  r.c.distance = p->distance;
  r.c.maxNeighborhood = p->maxNeighborhood;
  r.c.maxPruneCount = p->maxPruneCount;
  r.c.max_locations = p->max_locations;
  r.c.genome = genome ? genome : "";
  r.c.outfile = outfile ? outfile : "";
  std::vector<std::string> lines;
  std::string cur;
  for (const char* q = fasta; *q; ++q) {
    if (*q == '\n') {
      lines.push_back(cur);
      cur.clear();
    } else cur.push_back(*q);
  }
  if (!cur.empty()) lines.push_back(cur);
  int code = 0;
  std::string js = r.run(lines, code);
  if (rc) *rc = code;
  return dup_out(js, json_len);
}

struct orc_padlock_params {
  int32_t json, hamming, probe_mode, overlapping, compute_all, input_fasta, absent;
  uint32_t distance, armlen, tmdiff;
  double gcmin, gcmax;
};

static std::vector<std::string> split_lines(const char* text) {  // std::getline over the decompressed file
  std::vector<std::string> lines;
  std::string cur;
  for (const char* q = text; *q; ++q) {
    if (*q == '\n') {
      lines.push_back(cur);
      cur.clear();
    } else cur.push_back(*q);
  }
  if (!cur.empty()) lines.push_back(cur);
  return lines;
}

// padlock.h:147-531 (+ gtf.h).  strs = {ucscDB, anchor, spacerleft, spacerright, feature, idname, genome, infile, outfile,
// barcodes, gtf, jsonfile}; thal_fn = oracle/_ref's ref_thal, initialised by the caller.  Returns the TSV; *json_out the
// (uncompressed) JSON, *err_out what the reference writes to std::cerr.
char* orc_padlock(void* h, const orc_padlock_params* p, const char* const* strs, const char* const* genes, uint32_t ngenes,
                  const char* const* chrname, const char* const* chrseq, uint32_t nchr, const char* gtf_text, const char* bar_text,
                  void* thal_fn, int* rc, char** json_out, char** err_out) {
  PadlockRun r;
  r.fm = &((Handle*)h)->csa;
  r.thal = (ThalFn)thal_fn;
  r.c.json = p->json != 0;
  r.c.indel = !p->hamming;
  r.c.armMode = !p->probe_mode;
  r.c.overlapping = p->overlapping != 0;
  r.c.computeAll = p->compute_all != 0;
  r.c.inputFasta = p->input_fasta != 0;
  r.c.absent = p->absent != 0;
  r.c.distance = p->distance;
  r.c.armlen = p->armlen;
  r.c.tmdiff = p->tmdiff;
  r.c.mingcth = p->gcmin;
  r.c.maxgcth = p->gcmax;
  r.c.ucscDB = strs[0];
  r.c.anchor = strs[1];
  r.c.spacerleft = strs[2];
  r.c.spacerright = strs[3];
  r.c.feature = strs[4];
  r.c.idname = strs[5];
  r.c.genome = strs[6];
  r.c.infile = strs[7];
  r.c.outfile = strs[8];
  r.c.barcodes = strs[9];
  r.c.gtf = strs[10];
  r.c.jsonfile = strs[11];
  for (uint32_t i = 0; i < ngenes; ++i) r.c.geneset.insert(genes[i]);
  for (uint32_t i = 0; i < nchr; ++i) {
    r.chrname.push_back(chrname[i]);
    r.chrseq.push_back(chrseq[i]);
  }
  std::string tsv, json;
  int code = r.run(split_lines(gtf_text), split_lines(bar_text), tsv, json);
  if (rc) *rc = code;
  if (json_out) *json_out = dup_out(json, nullptr);
  if (err_out) *err_out = dup_out(r.err, nullptr);
  return dup_out(tsv, nullptr);
}

}  // extern "C"
