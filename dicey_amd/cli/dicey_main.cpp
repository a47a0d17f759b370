// `dicey` host binary for the MI355X search path: keeps the command line and the JSON of the reference's
// `dicey hunt` (reference src/dicey.cpp:35-76 dispatch, src/hunter.h:177-447) and calls the HIP kernels through the
// C ABI of libdiceygpu.so.  `dicey index` is provided as well (src/index.h:34-141) so that a genome can be prepared
// without the reference binary; `search` and `padlock` (padlock.cpp) ride on the same library; chop/mappability are out of scope.
//
// Host-side work that must stay on the host for bit-identical output: the hit vector arrives in the
// reference's push order and is sorted with libstdc++ std::sort under the reference's comparator (hunter.h:63-65,440).
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>
#include <string_view>
#include <thread>
#include <functional>
#include <fcntl.h>
#include <sys/stat.h>
#include <atomic>
#include <vector>

#include "../../include/dicey_gpu.h"
#include <dlfcn.h>
#include <unistd.h>
#include <chrono>

#include "../../include/dicey_gather.h"
#include "cli_common.hpp"
#include "dtoa.hpp"

int padlock_main(int argc, char** argv);  // padlock.cpp

namespace {

struct DnaHit {  // hunter.h:53-66
  int32_t score;
  uint32_t chr, start;
  char strand;
  std::string refalign, queryalign;
  bool operator<(const DnaHit& b) const {
    return ((score > b.score) || ((score == b.score) && (chr < b.chr)) || ((score == b.score) && (chr == b.chr) && (start < b.start)));
  }
};

inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline bool quick_exit_on() {  // DICEY_NO_QUICK_EXIT: close the index and let the HIP runtime unwind (leak checkers, debugging)
  static const bool on = std::getenv("DICEY_NO_QUICK_EXIT") == nullptr;
  return on;
}
inline bool timing_on() {
  static const bool on = std::getenv("DICEY_TIMING") != nullptr;
  return on;
}

// one query of the input: name and sequence as views of the input file's bytes (r06: 10 M records as pairs of std::string were 20 M
// heap allocations, half a second of a 3 s run); a sequence that spans several lines is joined in `owned`
struct Query {
  std::string_view first, second;
};

struct Config {
  std::string genome, outfile, input;
  bool has_outfile = false, hamming = false, forward = false, help = false;
  uint64_t max_locations = 1000;
  uint32_t max_neighborhood = 10000, distance = 1;
};

// hunter.h:99-160 (r04: appended in place — at 10 M queries per run the temporaries of the first form were the run time)
void hunt_json_append(std::string& o, const Config& c, uint32_t distance, std::string_view sequence, std::string_view qname,
                      const std::vector<std::string>& seqname, const std::vector<DnaHit>& ht, const std::vector<std::string>& msg) {
  o += "{\"errors\": [";
  bool errors = false;
  for (size_t i = 0; i < msg.size(); ++i) {
    bool err = msg[i].compare(0, 5, "Error") == 0;
    errors = errors || err;
    if (i) o.push_back(',');
    o += "{\"title\":";
    jstr_append(o, msg[i]);
    o += ",\"type\":";
    o += err ? "\"error\"" : "\"warning\"";
    o.push_back('}');
  }
  o.push_back(']');
  if (!errors) {
    o += ",\"meta\":{\"distance\":";
    uint_append(o, distance);
    o += ",\"forwardonly\":";
    o += c.forward ? "true" : "false";
    o += ",\"genome\":";
    jstr_append(o, c.genome);
    o += ",\"hamming\":";
    o += c.hamming ? "true" : "false";
    o += ",\"maxmatches\":";
    uint_append(o, c.max_locations);
    if (!qname.empty()) {
      o += ",\"name\":";
      jstr_append(o, qname.data(), qname.size());
    }
    o += ",\"outfile\":";
    jstr_append(o, c.outfile);
    o += ",\"sequence\":";
    jstr_append(o, sequence.data(), sequence.size());
    o += ",\"subcommand\":\"hunt\",\"version\":\"";
    o += kVersion;
    o += "\"},\"data\":[";
    uint32_t oldchr = 999999, oldstart = 0;
    bool first = true;
    for (const DnaHit& h : ht) {
      if (oldchr != h.chr || oldstart != h.start) {
        if (!first) o.push_back(',');
        first = false;
        uint32_t nuc = 0;
        for (char ch : h.refalign) nuc += ch != '-';
        o += "{\"chr\":";
        jstr_append(o, seqname[h.chr]);
        o += ",\"distance\":";
        uint_append(o, (uint64_t)std::abs(h.score));
        o += ",\"end\":";
        uint_append(o, (uint32_t)(h.start + nuc - 1));
        o += ",\"queryalign\":";
        jstr_append(o, h.queryalign);
        o += ",\"refalign\":";
        jstr_append(o, h.refalign);
        o += ",\"start\":";
        uint_append(o, h.start);
        o += ",\"strand\":\"";
        o.push_back(h.strand);
        o += "\"}";
      }
      oldchr = h.chr;
      oldstart = h.start;
    }
    o.push_back(']');
  }
  o += "}\n";
}
std::string hunt_json(const Config& c, uint32_t distance, std::string_view sequence, std::string_view qname,
                      const std::vector<std::string>& seqname, const std::vector<DnaHit>& ht, const std::vector<std::string>& msg) {
  std::string o;
  hunt_json_append(o, c, distance, sequence, qname, seqname, ht, msg);
  return o;
}

void emit(const Config& c, const std::string& json) {
  if (c.has_outfile) append_gzip_member(c.outfile, json);
  else {
    std::fwrite(json.data(), 1, json.size(), stdout);
    std::fflush(stdout);  // std::endl in the reference
  }
}

// ------------------------------------------------------------------------------------------------ options
const OptSpec kHuntOpts[] = {{"help", '?', false}, {"genome", 'g', true}, {"outfile", 'o', true}, {"maxmatches", 'm', true},
                             {"maxNeighborhood", 'x', true}, {"distance", 'd', true}, {"hamming", 'n', false}, {"forward", 'f', false},
                             {"input-file", 0, true}};
void hunt_usage(const char* sub) {
  std::cout << "Usage: dicey " << sub << " [OPTIONS] -g Danio_rerio.fa.gz CATTACTAACATCAGT" << std::endl;
  std::cout << "       dicey " << sub << " [OPTIONS] -g Danio_rerio.fa.gz sequences.fasta" << std::endl;
  std::cout << "Generic options:\n"
               "  -? [ --help ]                      show help message\n"
               "  -g [ --genome ] arg                genome file\n"
               "  -o [ --outfile ] arg               gzipped output file\n"
               "  -m [ --maxmatches ] arg (=1000)    max. number of matches\n"
               "  -x [ --maxNeighborhood ] arg (=10000)\n"
               "                                     max. neighborhood size\n"
               "  -d [ --distance ] arg (=1)         neighborhood distance\n"
               "  -n [ --hamming ]                   use hamming neighborhood instead of edit \n"
               "                                     distance\n"
               "  -f [ --forward ]                   only forward matches\n"
               "\n";
}

int device_from_env() {
  const char* e = std::getenv("DICEY_DEVICE");
  return e ? std::atoi(e) : 0;
}
// DICEY_DEVICES=0,1,2,...: the GPUs a batch is sharded over (a device may be listed more than once: one replica each)
std::vector<int> devices_from_env() {
  std::vector<int> out;
  if (const char* e = std::getenv("DICEY_DEVICES")) {
    std::string tok;
    for (const char* q = e;; ++q) {
      if (*q == ',' || *q == 0) {
        if (!tok.empty()) out.push_back(std::atoi(tok.c_str()));
        tok.clear();
        if (!*q) break;
      } else tok.push_back(*q);
    }
  }
  if (out.empty()) out.push_back(device_from_env());
  return out;
}

// ---------------------------------------------------------------------------- one process per GPU (DICEY_RANKS, hunter())
// libdiceygather.so is loaded only in this mode (it brings RCCL into the process): next to the binary, like libdiceygpu.so.
struct GatherApi {
  void* so = nullptr;
  int (*unique_id)(uint8_t*) = nullptr;
  int (*open)(const uint8_t*, int, int, int, uint64_t, int, dg_comm**) = nullptr;
  int (*open_tcp)(int, int, int, uint64_t, int, dg_comm**) = nullptr;
  int (*close)(dg_comm*) = nullptr;
  int (*submit)(dg_comm*, void*, const void*, uint64_t) = nullptr;
  int (*finish)(dg_comm*, uint64_t*, uint64_t*) = nullptr;
  int (*last_to_host)(dg_comm*, int, void*, uint64_t, uint64_t*) = nullptr;
  int (*max_u64)(dg_comm*, uint64_t, uint64_t*) = nullptr;
  const char* (*last_error)() = nullptr;
  bool load(std::string& err) {
    char exe[4096];
    const ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
    std::string dir = n > 0 ? std::string(exe, (size_t)n) : std::string("./dicey");
    dir = dir.substr(0, dir.find_last_of('/') + 1);
    so = dlopen((dir + "libdiceygather.so").c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!so) {
      err = std::string("libdiceygather.so: ") + dlerror();
      return false;
    }
#define DG_SYM(field, name)                                     \
    field = reinterpret_cast<decltype(field)>(dlsym(so, name)); \
    if (!field) {                                               \
      err = std::string("libdiceygather.so lacks ") + name;     \
      return false;                                             \
    }
    DG_SYM(unique_id, "dg_comm_unique_id")
    DG_SYM(open, "dg_comm_open")
    DG_SYM(open_tcp, "dg_comm_open_tcp")
    DG_SYM(close, "dg_comm_close")
    DG_SYM(submit, "dg_gather_submit")
    DG_SYM(finish, "dg_gather_finish")
    DG_SYM(last_to_host, "dg_gather_last_to_host")
    DG_SYM(max_u64, "dg_comm_max_u64")
    DG_SYM(last_error, "dg_gather_last_error")
#undef DG_SYM
    return true;
  }
};

// stdout behind a writer thread (r06): the formatting threads of chunk k + 1 work while chunk k's buffers are written, in order.
// A 10 M-query run writes 5.2 GB of JSON; one write() stream into a file runs at ~2.8 GB/s (page allocation + copy), which was
// spent AFTER each chunk's formatting.  (Measured and dropped: the formatting threads copying their lines into a shared mapping of
// the file — page faults of a mapping are slower than write() and do not scale — and parallel pwrite(), which serialises on the
// inode: 1.7 / 3.1 against 2.8 GB/s on 8 cores.)  At most 512 MB wait in the queue.
class AsyncWriter {
 public:
  ~AsyncWriter() { finish(); }
  void push(std::string&& blob) {
    if (blob.empty()) return;
    std::unique_lock<std::mutex> lk(mu_);
    if (!th_.joinable()) th_ = std::thread([this] { run(); });
    cv_room_.wait(lk, [this] { return queued_ < kMax; });
    queued_ += blob.size();
    q_.push_back(std::move(blob));
    cv_work_.notify_one();
  }
  void finish() {  // everything handed over so far is on stdout when this returns
    {
      std::unique_lock<std::mutex> lk(mu_);
      done_ = true;
      cv_work_.notify_one();
    }
    if (th_.joinable()) th_.join();
    std::unique_lock<std::mutex> lk(mu_);
    done_ = false;
    std::fflush(stdout);
  }

 private:
  void run() {
    for (;;) {
      std::string blob;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_work_.wait(lk, [this] { return done_ || !q_.empty(); });
        if (q_.empty()) return;
        blob = std::move(q_.front());
        q_.pop_front();
      }
      std::fwrite(blob.data(), 1, blob.size(), stdout);
      {
        std::unique_lock<std::mutex> lk(mu_);
        queued_ -= blob.size();
        cv_room_.notify_all();
      }
    }
  }
  static constexpr size_t kMax = 512u << 20;
  std::mutex mu_;
  std::condition_variable cv_work_, cv_room_;
  std::deque<std::string> q_;
  size_t queued_ = 0;
  bool done_ = false;
  std::thread th_;
};

template <class FormatFn>
int hunt_ranks(const Config& c, int N, int rank, dg_index* ix, int device, const std::vector<Query>& queries,
               const std::vector<uint32_t>& seqlen, const dg_hunt_params& hp, FormatFn& format_chunk_to) {
  auto die = [&](const std::string& m) {
    std::cerr << "dicey (rank " << rank << " of " << N << "): " << m << std::endl;
    return 2;
  };
  if (rank < 0 || rank >= N) return die("DICEY_RANK outside [0, DICEY_RANKS)");
  GatherApi G;
  std::string err;
  if (!G.load(err)) return die(err);
  const uint64_t CAP = 64ull << 20;  // bytes per transfer; a chunk's block travels in as many pieces as the largest rank's needs
  dg_comm* comm = nullptr;
  // DICEY_COMM_TCP=<port>: the library's host-memory test transport (two ranks on the one GPU of a test box; RCCL does not
  // take two ranks on one device).  The product form is RCCL: rank 0 writes the communicator id to DICEY_COMM_FILE.
  const char* tcp = std::getenv("DICEY_COMM_TCP");
  if (tcp) {
    if (G.open_tcp(std::atoi(tcp), N, rank, CAP, 0, &comm) != 0) return die(G.last_error());
  } else {
    const char* cf = std::getenv("DICEY_COMM_FILE");
    // the file holds the communicator id followed by a 16-byte launch token (DICEY_COMM_TOKEN, zero-padded; empty = all zero):
    // a file another launch left behind does not carry this launch's token and is not taken.  Rank 0 removes whatever lies at the
    // path before it writes, and removes its own file once the communicator stands (every rank has read it by then).
    uint8_t id[DG_COMM_ID_BYTES], token[16] = {0};
    if (const char* tk = std::getenv("DICEY_COMM_TOKEN")) std::memcpy(token, tk, std::min<size_t>(std::strlen(tk), sizeof token));
    if (N > 1 && !cf) return die("DICEY_COMM_FILE (where rank 0 leaves the communicator id) is not set");
    if (rank == 0) {
      if (cf) std::remove(cf);
      if (G.unique_id(id) != 0) return die(G.last_error());
      if (cf) {
        const std::string tmp = std::string(cf) + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id || std::fwrite(token, 1, sizeof token, f) != sizeof token)
          return die("cannot write " + tmp);
        std::fclose(f);
        if (std::rename(tmp.c_str(), cf) != 0) return die(std::string("cannot create ") + cf);
      }
    } else {
      bool got = false;
      for (int attempt = 0; attempt < 1200 && !got; ++attempt) {  // the ranks start together; rank 0 writes within milliseconds
        if (FILE* f = std::fopen(cf, "rb")) {
          uint8_t tk[sizeof token];
          got = std::fread(id, 1, sizeof id, f) == sizeof id && std::fread(tk, 1, sizeof tk, f) == sizeof tk &&
                std::memcmp(tk, token, sizeof tk) == 0;
          std::fclose(f);
        }
        if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(50));
      }
      if (!got) return die(std::string("no communicator id of this launch in ") + cf);
    }
    // RCCL greets with a version banner on stdout (rank 0, communicator creation); stdout is the JSON stream (hunter.h:160-175), so
    // the banner goes to stderr: file descriptor 1 points there while the communicator is created
    std::fflush(stdout);
    const int saved_out = dup(1);
    if (saved_out >= 0) dup2(2, 1);
    const int orc = G.open(id, N, rank, device, CAP, 0, &comm);
    std::fflush(stdout);
    if (saved_out >= 0) {
      dup2(saved_out, 1);
      close(saved_out);
    }
    if (orc != 0) return die(G.last_error());
    if (rank == 0 && cf) std::remove(cf);  // the communicator stands: every rank has read the id
  }
  const size_t nq = queries.size(), per = (nq + N - 1) / N;
  const size_t CH = 1u << 17;
  const size_t nchunks = std::max<size_t>(1, (per + CH - 1) / CH);  // the same on every rank: collectives stay in step
  auto range_of = [&](int r, size_t k, size_t& q0, size_t& q1) {
    const size_t s0 = std::min(nq, (size_t)r * per), s1 = std::min(nq, s0 + per);
    q0 = std::min(s1, s0 + k * CH);
    q1 = std::min(s1, q0 + CH);
  };
  std::vector<uint64_t> seq_start(seqlen.size() + 1, 0);
  for (size_t i = 0; i < seqlen.size(); ++i) seq_start[i + 1] = seq_start[i] + seqlen[i];
  // rank 0 keeps every (rank, chunk, piece) block on the host until everything has arrived, then formats in query order.
  // A chunk normally is ONE piece.  When a rank's library refuses its chunk's size (DG_ELIMIT: capped neighbourhoods beyond the
  // host budget — what run_slice answers by halves) the ranks agree, through the collective that carries the byte counts, to
  // cut the chunk into twice as many equal pieces ON EVERY RANK and start it again: the split is a function of (range, pieces)
  // alone, so rank 0 can rebuild every rank's pieces and the collectives stay in step.  Any other failure travels the same way
  // and ends every rank (one rank leaving alone left its peers blocked in the next collective).
  auto piece_of = [](size_t q0, size_t q1, uint64_t parts, uint64_t pc, size_t& a, size_t& b) {
    const size_t n = q1 - q0;
    a = q0 + (size_t)((unsigned __int128)n * pc / parts);
    b = q0 + (size_t)((unsigned __int128)n * (pc + 1) / parts);
  };
  constexpr uint64_t ST_SHIFT = 60, ST_SPLIT = 1, ST_FAIL = 2;
  std::vector<std::vector<std::vector<std::vector<uint8_t>>>> got(rank == 0 ? N : 0, std::vector<std::vector<std::vector<uint8_t>>>(nchunks));
  std::vector<uint64_t> parts_of(nchunks, 1);
  int rc_all = 0;
  for (size_t k = 0; k < nchunks; ++k) {
    size_t q0, q1;
    range_of(rank, k, q0, q1);
    for (uint64_t parts = 1;;) {
      bool again = false;
      if (rank == 0)
        for (int r = 0; r < N; ++r) got[r][k].assign(parts, std::vector<uint8_t>());
      for (uint64_t piece = 0; piece < parts && !again; ++piece) {
        size_t a, b;
        piece_of(q0, q1, parts, piece, a, b);
        dg_hunt_result* R = nullptr;
        uint64_t status = 0;
        if (b > a) {
          std::string qb;
          std::vector<uint64_t> off(b - a + 1, 0);
          size_t longest = 0;
          for (size_t i = 0; i < b - a; ++i) {
            qb += queries[a + i].second;
            off[i + 1] = qb.size();
            longest = std::max(longest, queries[a + i].second.size());
          }
          dg_hunt_params cp = hp;
          cp.max_query_len = (uint32_t)std::min<size_t>(longest, 0xFFFFFFu);
          cp.flags = DG_HUNT_COMPACT;
          const int rc = dg_hunt(ix, &cp, seqlen.data(), (uint32_t)seqlen.size(), (const uint8_t*)qb.data(), off.data(), b - a, &R);
          if (rc == DG_ELIMIT && b - a > 1) status = ST_SPLIT;
          else if (rc != DG_OK) {
            die(dg_last_error());  // outside the supported envelope: the whole job fails, never a partial answer
            status = ST_FAIL;
          }
          if (rc != DG_OK) R = nullptr;
        }
        const uint64_t bytes = R ? R->d_block_bytes : 0;
        uint64_t most = 0;
        if (G.max_u64(comm, (status << ST_SHIFT) | bytes, &most) != 0) return die(G.last_error());
        if ((most >> ST_SHIFT) >= ST_FAIL) {  // some rank cannot answer: every rank leaves, none waits for a peer that is gone
          if (R) dg_hunt_result_free(R);
          G.close(comm);
          return status == ST_FAIL ? 2 : die("another rank failed; no output");
        }
        if ((most >> ST_SHIFT) == ST_SPLIT) {
          if (R) dg_hunt_result_free(R);
          parts *= 2;
          again = true;
          break;
        }
        const uint64_t pieces = std::max<uint64_t>(1, (most + CAP - 1) / CAP);
        for (uint64_t pc = 0; pc < pieces; ++pc) {
          const uint64_t b0 = std::min(bytes, pc * CAP), b1 = std::min(bytes, b0 + CAP);
          // RCCL: out of HBM, staged on the stream the batch ran on; the TCP test transport takes the fetched copy of the same bytes
          const uint8_t* src = !R ? nullptr : tcp ? (const uint8_t*)R->qinfo - 4 * (b - a) : (const uint8_t*)R->d_block;
          if (G.submit(comm, (R && !tcp) ? R->stream : nullptr, src ? src + b0 : nullptr, b1 - b0) != 0) return die(G.last_error());
          if (G.finish(comm, nullptr, nullptr) != 0) return die(G.last_error());
          if (rank == 0)
            for (int r = 0; r < N; ++r) {
              uint64_t n = 0;
              if (G.last_to_host(comm, r, nullptr, 0, &n) != 0 && n == 0) return die(G.last_error());
              std::vector<uint8_t>& dst = got[r][k][piece];
              const size_t at = dst.size();
              dst.resize(at + n);
              if (n && G.last_to_host(comm, r, dst.data() + at, n, &n) != 0) return die(G.last_error());
            }
        }
        if (R) dg_hunt_result_free(R);
      }
      if (!again) {
        parts_of[k] = parts;
        break;
      }
    }
  }
  G.close(comm);
  if (rank != 0) return 0;
  std::function<void(std::string&&)> bulk;
  if (!c.has_outfile)
    bulk = [&](std::string&& blob) {
      std::fwrite(blob.data(), 1, blob.size(), stdout);
      std::fflush(stdout);
    };
  auto sink = [&](size_t, std::string&& js) { emit(c, js); };
  for (int r = 0; r < N && !rc_all; ++r)
    for (size_t k = 0; k < nchunks && !rc_all; ++k)
      for (uint64_t piece = 0; piece < parts_of[k] && !rc_all; ++piece) {
        size_t c0, c1, q0, q1;
        range_of(r, k, c0, c1);
        piece_of(c0, c1, parts_of[k], piece, q0, q1);
        const size_t n = q1 - q0;
        if (!n) continue;
        const std::vector<uint8_t>& blk = got[r][k][piece];
        if (blk.size() < 8 * n) {
          rc_all = die("rank " + std::to_string(r) + " sent " + std::to_string(blk.size()) + " bytes for " + std::to_string(n) + " queries");
          break;
        }
        // the block as a result: [hit counts | query words | records]
        dg_hunt_result S{};
        std::vector<uint64_t> hit_off(n + 1, 0);
        const uint32_t* qh = (const uint32_t*)blk.data();
        for (size_t i = 0; i < n; ++i) hit_off[i + 1] = hit_off[i] + qh[i];
        S.nq = n;
        S.nhits = hit_off[n];
        const uint64_t rec_bytes = blk.size() - 8 * n;
        if (S.nhits && (rec_bytes % (4 * S.nhits) != 0 || rec_bytes / (4 * S.nhits) < 2)) {
          rc_all = die("rank " + std::to_string(r) + ": " + std::to_string(rec_bytes) + " record bytes for " + std::to_string(S.nhits) + " hits");
          break;
        }
        S.ops_per_hit = S.nhits ? (uint32_t)(rec_bytes / (4 * S.nhits)) - 2 : 0;
        S.hit_off = hit_off.data();
        S.qinfo = const_cast<uint32_t*>(qh) + n;
        S.chits = const_cast<uint32_t*>(qh) + 2 * n;
        S.compact = 1;
        S.nseq = (uint32_t)seqlen.size();
        S.seq_start = seq_start.data();
        format_chunk_to(&S, q0, n, sink, bulk);
      }
  return rc_all;
}

// ------------------------------------------------------------------------------------------------ hunt (hunter.h:177-447)
int hunter(int argc, char** argv) {
  Config c;
  Parsed p = parse_options(argc, argv, kHuntOpts, sizeof kHuntOpts / sizeof kHuntOpts[0]);
  if (!p.error.empty()) {  // program_options throws; the reference does not catch -> terminate
    std::cerr << "terminate called after throwing an instance of 'boost::program_options::error'\n  what():  " << p.error << std::endl;
    std::abort();
  }
  bool have_genome = false, have_input = false;
  for (auto& kv : p.kv) {
    const std::string& k = kv.first;
    if (k == "help") c.help = true;
    else if (k == "genome") { c.genome = kv.second; have_genome = true; }
    else if (k == "outfile") { c.outfile = kv.second; c.has_outfile = true; }
    else if (k == "maxmatches") c.max_locations = std::strtoull(kv.second.c_str(), nullptr, 10);
    else if (k == "maxNeighborhood") c.max_neighborhood = (uint32_t)std::strtoul(kv.second.c_str(), nullptr, 10);
    else if (k == "distance") c.distance = (uint32_t)std::strtoul(kv.second.c_str(), nullptr, 10);
    else if (k == "hamming") c.hamming = true;
    else if (k == "forward") c.forward = true;
    else if (k == "input-file") { c.input = kv.second; have_input = true; }
  }
  if (!p.positional.empty()) {
    c.input = p.positional.back();
    have_input = true;
  }
  if (c.help || !have_input || !have_genome) {
    hunt_usage(argv[0]);
    return -1;
  }
  std::vector<DnaHit> none;
  std::vector<std::string> msg;
  std::vector<uint32_t> seqlen;
  std::vector<std::string> seqname;
  if (c.has_outfile) {  // hunter.h:232-235
    std::ofstream trunc(c.outfile.c_str(), std::ios_base::out | std::ios_base::binary | std::ios_base::trunc);
  }
  if (!file_nonempty(c.genome)) {
    msg.push_back("Error: Genome does not exist!");
    emit(c, hunt_json(c, c.distance, c.input, "", seqname, none, msg));
    return 1;
  }
  if (!seq_len_name(c.genome, seqlen, seqname)) {
    msg.push_back("Error: Could not retrieve sequence lengths!");
    emit(c, hunt_json(c, c.distance, c.input, "", seqname, none, msg));
    return 1;
  }
  std::string index_file = strip_last_extension(c.genome) + ".fm9";
  // hunter.h:262-287: a FASTA file of queries, or the literal sequence (read before the index is opened: the shard plan
  // needs the count; a malformed input is still reported after a failing index, as in the reference)
  std::vector<Query> queries;
  std::string qfile;                 // the input file's bytes: names and sequences are views of it
  std::deque<std::string> owned;     // sequences that span several lines, joined
  bool bad_fasta = false;
  const double t_read0 = now_ms();
  // (r06: on a thread of its own beside the index open — 0.3 s for a 10 M-record file; joined before anything looks at the queries)
  std::thread reader([&]() {
  if (is_regular(c.input)) {
    if (!is_fasta(c.input)) bad_fasta = true;
    else {
      // hunter.h:272-283 over the whole file in memory (10 M records: std::getline per line was seconds)
      if (FILE* f = std::fopen(c.input.c_str(), "rb")) {
        std::fseek(f, 0, SEEK_END);
        const long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        qfile.resize(sz > 0 ? (size_t)sz : 0);
        if (sz > 0 && std::fread(&qfile[0], 1, (size_t)sz, f) != (size_t)sz) qfile.clear();
        std::fclose(f);
      }
      queries.reserve(qfile.size() / 28 + 16);
      // a record = a name line and the non-empty lines up to the next '>' (std::getline keeps a '\r', so does this); the reference
      // keeps a record when name and sequence are both non-empty (hunter.h:276-283)
      std::string_view name, seq;
      bool have = false, joined = false;
      auto flush = [&]() {
        if (have && !name.empty() && !seq.empty()) queries.push_back(Query{name, seq});
        else if (joined) owned.pop_back();
      };
      const char* p0 = qfile.data();
      const char* const end = p0 + qfile.size();
      while (p0 < end) {
        const char* nl = (const char*)std::memchr(p0, '\n', (size_t)(end - p0));
        const char* e = nl ? nl : end;
        if (e > p0) {  // (empty lines are skipped)
          if (*p0 == '>') {
            flush();
            name = std::string_view(p0 + 1, (size_t)(e - p0 - 1));
            seq = std::string_view();
            have = true;
            joined = false;
          } else if (seq.empty() && !joined) seq = std::string_view(p0, (size_t)(e - p0));
          else {  // a second sequence line: the record's sequence moves into a string of its own
            if (!joined) {
              owned.emplace_back(seq);
              joined = true;
            }
            owned.back().append(p0, e);
            seq = owned.back();
          }
        }
        p0 = nl ? nl + 1 : end;
      }
      flush();
    }
  } else queries.push_back(Query{std::string_view(), std::string_view(c.input)});
  if (timing_on()) std::fprintf(stderr, "dicey timing: %-28s %8.1f ms\n", "input read + records", now_ms() - t_read0);
  });
  struct Joiner {
    std::thread& t;
    ~Joiner() { if (t.joinable()) t.join(); }
  } reader_joiner{reader};

  // the K-mer jump table (up to 137 GB, ~1.5 s to derive) only pays off for large batches: a literal sequence or a
  // small FASTA is answered from the Occ blocks alone
  // a process that opens the index for one input keeps the table compact (DG_OPEN_COMPACT): the 137 GB form saves microseconds
  // per batch and costs seconds to allocate when the driver still has to wipe memory another process released
  // (DG_OPEN_COMPACT since ABI 7: without the layouts that only pay for a resident index — the suffix array with context and its
  //  prefix levels are 40 GB, 0.36 s to derive and a second of driver wipe at exit, against 0.3 ms per 100 000 queries of a repeat-rich batch)
  uint32_t open_flags = (std::getenv("DICEY_FULL_TABLE") ? DG_OPEN_BIG_TABLE : DG_OPEN_DEFAULT) | (std::getenv("DICEY_RESIDENT_LAYOUTS") ? 0u : DG_OPEN_COMPACT);
  {
    struct stat ist;
    const bool is_file = stat(c.input.c_str(), &ist) == 0 && S_ISREG(ist.st_mode);
    if (!(is_file && ist.st_size > (1 << 20)) && !std::getenv("DICEY_KMER_K")) open_flags |= DG_OPEN_NO_KMER_TABLE;
    // the preceding-characters array (0.19 s to derive) saves a fifth of the distance-1 search kernel and a twentieth of the
    // distance-2 one: 0.3 ms per million queries at distance 1, 3 ms at distance 2 — left out unless the input is large enough to earn it back
    if ((open_flags & DG_OPEN_COMPACT) && !(is_file && ist.st_size > (c.distance >= 2 ? (1ll << 30) : (8ll << 30)))) open_flags |= DG_OPEN_NO_PRE5;
  }
  // Queries are independent (hunter.h:291), so a batch shards over the GPUs of the node by contiguous ranges
  // (ceil(nq/G) each): DICEY_DEVICES=0,1,... gives one host thread and one full index replica per listed
  // device; results are written in query order.  One device (DICEY_DEVICE, default 0): results stream out chunk by chunk.
  std::vector<int> devices = devices_from_env();
  if (std::getenv("DICEY_RANKS")) {  // one process per GPU: this process's device (DICEY_DEVICE, else its rank)
    const int one = std::getenv("DICEY_DEVICE") ? device_from_env() : (std::getenv("DICEY_RANK") ? std::atoi(std::getenv("DICEY_RANK")) : 0);
    devices.assign(1, one);
  }
  if (devices.size() > 1) {  // the shard plan needs the count
    reader.join();
    if (devices.size() > queries.size()) devices.resize(std::max<size_t>(1, queries.size()));
  }
  const size_t G = devices.size();
  std::vector<dg_index*> handles(G, nullptr);
  std::vector<std::string> open_err(G);
  {
    std::vector<std::thread> pool;
    for (size_t g = 0; g < G; ++g)
      pool.emplace_back([&, g]() {
        if (dg_index_open(index_file.c_str(), devices[g], open_flags, &handles[g]) != DG_OK) open_err[g] = dg_last_error();
      });
    for (auto& t : pool) t.join();
  }
  if (reader.joinable()) reader.join();
  auto close_all = [&]() {
    const double t_close0 = now_ms();
    for (dg_index* h : handles)
      if (h) dg_index_close(h);
    if (timing_on()) std::fprintf(stderr, "dicey timing: %-28s %8.1f ms\n", "index close", now_ms() - t_close0);
  };
  for (size_t g = 0; g < G; ++g)
    if (!handles[g]) {
      std::cerr << "dicey: " << open_err[g] << std::endl;
      msg.push_back("Error: FM-Index cannot be loaded!");
      emit(c, hunt_json(c, c.distance, c.input, "", seqname, none, msg));
      close_all();
      return 1;
    }
  if (bad_fasta) {
    msg.push_back("Error: Input file is not in FASTA format!");
    emit(c, hunt_json(c, c.distance, c.input, "", seqname, none, msg));
    close_all();
    return 1;
  }

  dg_hunt_params hp{};
  hp.distance = c.distance;
  hp.hamming = c.hamming;
  hp.forward_only = c.forward;
  hp.max_locations = c.max_locations;
  hp.max_neighborhood = c.max_neighborhood;
  // one JSON line per query of a finished chunk, handed to the sink in query order
  auto format_chunk_to = [&](dg_hunt_result* R, size_t q0, size_t nq, const std::function<void(size_t, std::string&&)>& sink,
                             const std::function<void(std::string&&)>& bulk) {
    auto line_of = [&](size_t i) -> std::string {
      std::vector<std::string> m;
      std::vector<DnaHit> ht;
      const std::string_view qname = queries[q0 + i].first;
      // compact results (DG_HUNT_COMPACT): one word per query; the normalised sequence is formed here from the query's own bytes
      const uint32_t qw = R->qinfo[i], qfl = DG_QINFO_FLAGS(qw);
      if (qfl & DG_Q_TOO_SHORT) {
        m.push_back("Error: Input sequence is shorter than 10 nucleotides!");
        return hunt_json(c, c.distance, queries[q0 + i].second, qname, seqname, ht, m);
      }
      const std::string_view raw = queries[q0 + i].second;
      std::string seq(raw.size(), '\0');
      uint32_t nondna = 0;
      (void)dg_normalize_query((const uint8_t*)raw.data(), (uint32_t)raw.size(), (uint8_t*)seq.data(), &nondna);
      for (uint32_t k = 0; k < nondna; ++k) m.push_back("Warning: Non-DNA character in nucleotide sequence detected and replaced by 'N'!");
      if (qfl & DG_Q_DIST_ADJUSTED) m.push_back("Warning: Distance was adjusted to sequence length!");
      if (qfl & DG_Q_NBHD_EXCEEDED) {
        std::string x = std::to_string(c.max_neighborhood);
        m.push_back("Warning: Neighborhood size exceeds " + x + " candidates. Only first " + x + " neighbors are searched, results are likely incomplete!");
      }
      for (uint64_t h = R->hit_off[i]; h < R->hit_off[i + 1]; ++h) {
        // the hit from its compact record (dicey_gpu.h ABI 5: position, packed word, operation words) and its two rows from the
        // operation words — built here, on the formatting threads
        dg_hit H;
        const uint32_t* hops = nullptr;
        if (dg_chit_unpack(R, h, (uint32_t)i, &H, &hops) != DG_OK) {
          std::fprintf(stderr, "dicey hunt: %s\n", dg_last_error());
          std::abort();
        }
        std::string ra(H.aln_len, '\0'), qa(H.aln_len, '\0');
        if (dg_hit_rows(&H, hops, R->ops_per_hit, (const uint8_t*)seq.data(), (uint32_t)seq.size(), ra.data(), qa.data()) != DG_OK) {
          std::fprintf(stderr, "dicey hunt: %s\n", dg_last_error());
          std::abort();
        }
        ht.push_back(DnaHit{H.score, H.chr, H.start, (char)H.strand, std::move(ra), std::move(qa)});
      }
      if (qfl & DG_Q_MAX_MATCHES) {
        std::string x = std::to_string(c.max_locations);
        m.push_back("Warning: More than " + x + " matches found. Only first " + x + " matches are reported, results are likely incomplete!");
      }
      std::sort(ht.begin(), ht.end());  // hunter.h:440 — same comparator, same libstdc++ algorithm, same input order
      return hunt_json(c, DG_QINFO_DISTANCE(qw), seq, qname, seqname, ht, m);
    };
    unsigned nthr = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("DICEY_HOST_THREADS")) nthr = (unsigned)std::max(1, std::atoi(e));
    nthr = (unsigned)std::min<size_t>(std::min<unsigned>(nthr ? nthr : 1u, 64u), nq / 32);
    if (nthr < 1) nthr = 1;
    if (bulk) {
      // stdout: every formatting thread appends its lines to one buffer, the buffers go out in order with one write each (the
      // reference flushes after every line; the bytes are the same)
      std::vector<std::string> blobs(nthr);
      const size_t per_thr = (nq + nthr - 1) / nthr;
      auto work = [&](unsigned t) {
        std::string& o = blobs[t];
        const size_t i0 = t * per_thr, i1 = std::min(nq, i0 + per_thr);
        static std::atomic<size_t> per_line{700};  // bytes per line of the last buffer formatted (distance 2: ~8 KB): the next one is sized for it
        o.reserve((i1 > i0 ? i1 - i0 : 0) * per_line.load(std::memory_order_relaxed));
        for (size_t i = i0; i < i1; ++i) o += line_of(i);
        if (i1 > i0 + 16) per_line.store(o.size() / (i1 - i0) + o.size() / (i1 - i0) / 8 + 64, std::memory_order_relaxed);
      };
      if (nthr > 1) {
        std::vector<std::thread> fmt;
        for (unsigned t = 0; t < nthr; ++t) fmt.emplace_back(work, t);
        for (auto& th : fmt) th.join();
      } else work(0);
      for (std::string& o : blobs) bulk(std::move(o));  // (the writer thread's queue: the next chunk is formatted while these go out)
      return;
    }
    if (nthr > 1) {
      std::vector<std::string> lines(nq);
      std::vector<std::thread> fmt;
      const size_t per_thr = (nq + nthr - 1) / nthr;
      for (unsigned t = 0; t < nthr; ++t)
        fmt.emplace_back([&, t]() {
          for (size_t i = t * per_thr, e = std::min(nq, i + per_thr); i < e; ++i) lines[i] = line_of(i);
        });
      for (auto& th : fmt) th.join();
      for (size_t i = 0; i < nq; ++i) sink(q0 + i, std::move(lines[i]));
    } else {
      for (size_t i = 0; i < nq; ++i) sink(q0 + i, line_of(i));
    }
  };
  // queries [q0, q1) on one handle, chunk by chunk; sink(i, json line of query i) is called in query order
  auto run_slice = [&](dg_index* ix, size_t s0, size_t s1, const std::function<void(size_t, std::string&&)>& sink, std::string& err,
                       const std::function<void(std::string&&)>& bulk = nullptr) -> bool {
    auto pack = [&](size_t q0, size_t q1, std::string& qb, std::vector<uint64_t>& off) -> dg_hunt_params {
      qb.clear();
      off.assign(q1 - q0 + 1, 0);
      size_t longest = 0;
      for (size_t i = 0; i < q1 - q0; ++i) {
        qb += queries[q0 + i].second;
        off[i + 1] = qb.size();
        longest = std::max(longest, queries[q0 + i].second.size());
      }
      dg_hunt_params cp = hp;  // the chunk's own bound: the library sizes the batch without another pass over the offsets
      cp.max_query_len = (uint32_t)std::min<size_t>(longest, 0xFFFFFFu);
      cp.flags = DG_HUNT_COMPACT;
      return cp;
    };
    auto format_chunk = [&](dg_hunt_result* R, size_t q0, size_t nq) { format_chunk_to(R, q0, nq, sink, bulk); };
    // a chunk through the blocking call; a chunk whose capped neighbourhoods would not fit the library's host budget (DG_ELIMIT,
    // hunt.hip cap_scan: long primers at distance 2) is answered in halves — the reference answers that input too, slowly
    std::function<bool(dg_index*, size_t, size_t)> run_sync = [&](dg_index* hx, size_t q0, size_t q1) -> bool {
      std::string qb;
      std::vector<uint64_t> off;
      const dg_hunt_params cp = pack(q0, q1, qb, off);
      dg_hunt_result* R = nullptr;
      const int rc = dg_hunt(hx, &cp, seqlen.data(), (uint32_t)seqlen.size(), (const uint8_t*)qb.data(), off.data(), q1 - q0, &R);
      if (rc == DG_ELIMIT && q1 - q0 > 1) {
        const size_t mid = q0 + (q1 - q0) / 2;
        return run_sync(hx, q0, mid) && run_sync(hx, mid, q1);
      }
      if (rc != DG_OK) {
        err = dg_last_error();  // outside the supported envelope: say so, never guess
        return false;
      }
      format_chunk(R, q0, q1 - q0);
      dg_hunt_result_free(R);
      return true;
    };
    // Large inputs go chunk by chunk with TWO batches in flight (dg_hunt_submit / dg_hunt_wait on this handle and on a second
    // one that shares the resident index): while the host formats the lines of chunk k, chunk k+1 is on the GPU.
    // (distance >= 2: a query's line is ~8 KB on a 3 Gb genome — 131 072 of them are a gigabyte, twice what the writer's queue takes,
    //  and the formatting threads stood still while it drained; 16 384 keep both sides busy: 1 M queries 4.1 -> see DESIGN §9)
    const size_t CHUNK = c.distance >= 2 ? (1u << 14) : (s1 - s0) > (1u << 18) ? (1u << 17) : (1u << 20);
    std::vector<std::pair<size_t, size_t>> chunks;
    for (size_t q0 = s0; q0 < s1; q0 += CHUNK) chunks.emplace_back(q0, std::min(s1, q0 + CHUNK));
    if (chunks.size() < 2 || std::getenv("DICEY_NO_PIPELINE")) {
      for (auto& ch : chunks)
        if (!run_sync(ix, ch.first, ch.second)) return false;
      return true;
    }
    dg_index* second = nullptr;
    if (dg_index_share(ix, &second) != DG_OK) second = nullptr;
    dg_index* lanes[2] = {ix, second ? second : ix};
    const size_t depth = second ? 2 : 1;
    std::vector<dg_hunt_ticket*> tickets(chunks.size(), nullptr);
    bool ok = true;
    auto submit = [&](size_t k) -> bool {
      std::string qb;
      std::vector<uint64_t> off;
      const dg_hunt_params cp = pack(chunks[k].first, chunks[k].second, qb, off);
      if (dg_hunt_submit(lanes[k % depth], &cp, seqlen.data(), (uint32_t)seqlen.size(), (const uint8_t*)qb.data(), off.data(),
                         chunks[k].second - chunks[k].first, &tickets[k]) != DG_OK) {
        err = dg_last_error();
        return false;
      }
      return true;
    };
    // Invariant: chunk k runs on lane k % depth, and while everything is fine chunk k + depth is submitted as soon as chunk k's
    // lane is idle again — right after its wait() when the chunk was answered (the next one is on the GPU while this one is
    // formatted), or after the blocking halves when the library refused the chunk's size (DG_ELIMIT: the lane must be idle for
    // run_sync).  A missing ticket while `ok` holds is a bug of this loop and ends the run with an error, never with silence.
    size_t submitted = 0;
    auto top_up = [&]() {
      if (ok && submitted < chunks.size()) {
        ok = submit(submitted);
        if (ok) ++submitted;
      }
    };
    for (; submitted < std::min(depth, chunks.size()) && ok;) top_up();
    for (size_t k = 0; k < chunks.size(); ++k) {
      if (!tickets[k]) {
        if (ok) {
          err = "internal error: chunk " + std::to_string(k) + " of " + std::to_string(chunks.size()) + " was never submitted";
          ok = false;
        }
        continue;
      }
      dg_hunt_result* R = nullptr;
      const int rc = dg_hunt_wait(tickets[k], &R);
      tickets[k] = nullptr;
      if (!ok) {
        if (R) dg_hunt_result_free(R);
        continue;  // keep collecting what is in flight
      }
      if (rc == DG_OK) {
        top_up();  // the lane is free again: the next chunk starts before this one is formatted
        format_chunk(R, chunks[k].first, chunks[k].second - chunks[k].first);
        dg_hunt_result_free(R);
      } else if (rc == DG_ELIMIT) {
        if (R) dg_hunt_result_free(R);
        ok = run_sync(lanes[k % depth], chunks[k].first, chunks[k].second);  // that lane is idle: its wait() has returned and nothing new was submitted to it
        top_up();
      } else {
        err = dg_last_error();
        ok = false;
        if (R) dg_hunt_result_free(R);
      }
    }
    if (second) dg_index_close(second);
    return ok;
  };
  // DICEY_RANKS=N DICEY_RANK=r DICEY_COMM_FILE=<path>: ONE PROCESS PER GPU (BASELINE.json north_star: "the primer batch shards
  // embarrassingly across the 8 GPUs of one node with RCCL over xGMI only to gather hit lists").  Every rank reads the same input,
  // searches its contiguous range of the queries (ceil(nq/N) each, rank order = query order) on its own index replica, and the ranks'
  // answers — per chunk the compact block [hit counts | query words | records], dg_hunt_result::d_block, straight out of HBM — are
  // gathered to rank 0 by libdiceygather.so (include/dicey_gather.h: exact-size ncclSend / ncclRecv), which formats and writes
  // every line in query order.  The reference is one process; what this replaces is the concatenation of the shards' results.
  if (const char* er = std::getenv("DICEY_RANKS")) {
    const int N = std::max(1, std::atoi(er));
    const int rank = std::getenv("DICEY_RANK") ? std::atoi(std::getenv("DICEY_RANK")) : 0;
    const int rr = hunt_ranks(c, N, rank, handles[0], devices[0], queries, seqlen, hp, format_chunk_to);
    close_all();
    return rr;
  }
  int rc_all = 0;
  if (G == 1) {
    std::string err;
    std::function<void(std::string&&)> bulk;
    AsyncWriter writer;
    if (!c.has_outfile)  // (the gzip outfile is one member per query, hunter.h:162-170: written line by line)
      bulk = [&](std::string&& blob) { writer.push(std::move(blob)); };
    const double t_run0 = now_ms();
    const bool ran = run_slice(handles[0], 0, queries.size(), [&](size_t, std::string&& js) { emit(c, js); }, err, bulk);
    writer.finish();
    if (timing_on()) std::fprintf(stderr, "dicey timing: %-28s %8.1f ms\n", "search + format + write", now_ms() - t_run0);
    if (!ran) {
      std::cerr << "dicey: " << err << std::endl;
      rc_all = 2;
    }
  } else {
    const size_t nq = queries.size(), per = (nq + G - 1) / G;
    std::vector<std::string> lines(nq);
    std::vector<std::string> errs(G);
    std::vector<char> ok(G, 1);
    std::vector<std::thread> pool;
    for (size_t g = 0; g < G; ++g)
      pool.emplace_back([&, g]() {
        const size_t s0 = std::min(nq, g * per), s1 = std::min(nq, s0 + per);
        ok[g] = run_slice(handles[g], s0, s1, [&](size_t i, std::string&& js) { lines[i] = std::move(js); }, errs[g]);
      });
    for (auto& t : pool) t.join();
    for (size_t g = 0; g < G && !rc_all; ++g) {  // query order; stop at the first shard that failed
      const size_t s0 = std::min(nq, g * per), s1 = std::min(nq, s0 + per);
      for (size_t i = s0; i < s1 && !lines[i].empty(); ++i) emit(c, lines[i]);
      if (!ok[g]) {
        std::cerr << "dicey: " << errs[g] << std::endl;
        rc_all = 2;
      }
    }
  }
  if (!quick_exit_on()) close_all();  // (main leaves with _exit right behind this)
  return rc_all;
}

// ------------------------------------------------------------------------------------------------ search (silica.h:208-651)
struct SearchConfig {
  std::string genome, outfile, infile, primer3Config = "./src/primer3_config/";
  bool has_outfile = false, hamming = false, pruneprimer = false, help = false;
  double cutTemp = 45.0, cutofPen = -1.0, penDiff = 0.6, penMis = 0.4, penLen = 0.001;
  double temp = 37.0, mv = 50.0, dv = 1.5, dna_conc = 50.0, dntp = 0.6;
  uint32_t maxProdSize = 15000, kmer = 15, distance = 1, maxNeighborhood = 10000, maxPruneCount = 0;
  uint64_t max_locations = 10000;
};
struct PrimerBind {  // silica.h:69-82
  uint32_t refIndex, pos, primerId;
  bool onFor;
  double temp, perfTemp;
  std::string genome;
  bool operator<(const PrimerBind& b) const { return (temp > b.temp); }
};
struct PcrProduct {  // silica.h:84-98
  uint32_t refIndex, leng, forPos, revPos, forId, revId;
  double forTemp, revTemp, penalty;
  bool operator<(const PcrProduct& b) const { return (penalty < b.penalty); }
};

// silica.h:100-187; amplicon sequences come from the index text (what faidx_fetch_seq + to_upper would return)
std::string search_json(const SearchConfig& c, dg_index* ix, const std::vector<uint32_t>& seqlen, const std::vector<std::string>& qn,
                        const std::vector<PrimerBind>& allp, const std::vector<PcrProduct>& pcr, const std::vector<std::string>& pName,
                        const std::vector<std::string>& pSeq, const std::vector<std::string>& msg) {
  std::string o = "{\"errors\": [";
  bool errors = false;
  for (size_t i = 0; i < msg.size(); ++i) {
    bool err = msg[i].compare(0, 5, "Error") == 0;
    errors = errors || err;
    if (i) o.push_back(',');
    o += "{\"title\":" + jstr(msg[i]) + ",\"type\":" + (err ? "\"error\"" : "\"warning\"") + "}";
  }
  o.push_back(']');
  if (!errors) {
    o += ",\"meta\":{\"distance\":" + std::to_string(c.distance) + ",\"genome\":" + jstr(c.genome);
    o += std::string(",\"hamming\":") + (c.hamming ? "true" : "false") + ",\"maxmatches\":" + std::to_string(c.max_locations);
    o += ",\"outfile\":" + jstr(c.outfile) + ",\"subcommand\":\"search\",\"version\":\"" + kVersion + "\"},";
    o += "\"data\":{\"primers\":[";
    for (size_t i = 0; i < allp.size(); ++i) {
      const PrimerBind& p = allp[i];
      if (i) o.push_back(',');
      o += "{\"Chrom\":" + jstr(qn[p.refIndex]) + ",\"End\":" + std::to_string((uint64_t)p.pos + pSeq[p.primerId].size());
      o += ",\"Genome\":" + jstr(p.genome) + ",\"Id\":" + std::to_string(i) + ",\"MatchTm\":" + dtoa::dump_double(p.perfTemp);
      o += ",\"Name\":" + jstr(pName[p.primerId]) + ",\"Ori\":" + (p.onFor ? "\"forward\"" : "\"reverse\"");
      o += ",\"Pos\":" + std::to_string(p.pos + 1) + ",\"Seq\":" + jstr(pSeq[p.primerId]) + ",\"Tm\":" + dtoa::dump_double(p.temp) + "}";
    }
    o += "],\"amplicons\":[";
    // amplicon sequences in one extract batch
    std::vector<uint64_t> lo, hi, off;
    uint64_t tot = 0;
    std::vector<char> has(pcr.size(), 0);
    for (size_t i = 0; i < pcr.size(); ++i) {
      const PcrProduct& a = pcr[i];
      uint64_t cstart = 0;
      for (uint32_t r = 0; r < a.refIndex; ++r) cstart += seqlen[r];
      uint64_t clen = seqlen[a.refIndex] - 1, b0 = a.forPos, e0 = (uint64_t)a.revPos + pSeq[a.revId].size() - 1;
      if (b0 < clen) {
        if (e0 >= clen) e0 = clen - 1;
        if (e0 >= b0) {
          has[i] = 1;
          lo.push_back(cstart + b0);
          hi.push_back(cstart + e0);
          off.push_back(tot);
          tot += e0 - b0 + 1;
        }
      }
    }
    std::string pool(tot, '\0');
    if (!lo.empty() && dg_extract(ix, lo.data(), hi.data(), lo.size(), (uint8_t*)&pool[0], off.data()) != DG_OK)
      std::cerr << "dicey: " << dg_last_error() << std::endl;
    size_t k = 0;
    for (size_t i = 0; i < pcr.size(); ++i) {
      const PcrProduct& a = pcr[i];
      std::string seqstr;
      if (has[i]) {
        seqstr = pool.substr(off[k], hi[k] - lo[k] + 1);
        ++k;
      }
      if (i) o.push_back(',');
      o += "{\"Chrom\":" + jstr(qn[a.refIndex]) + ",\"ForEnd\":" + std::to_string((uint64_t)a.forPos + pSeq[a.forId].size());
      o += ",\"ForName\":" + jstr(pName[a.forId]) + ",\"ForPos\":" + std::to_string(a.forPos + 1) + ",\"ForSeq\":" + jstr(pSeq[a.forId]);
      o += ",\"ForTm\":" + dtoa::dump_double(a.forTemp) + ",\"Id\":" + std::to_string(i) + ",\"Length\":" + std::to_string(a.leng);
      o += ",\"Penalty\":" + dtoa::dump_double(a.penalty) + ",\"RevEnd\":" + std::to_string((uint64_t)a.revPos + pSeq[a.revId].size());
      o += ",\"RevName\":" + jstr(pName[a.revId]) + ",\"RevPos\":" + std::to_string(a.revPos + 1) + ",\"RevSeq\":" + jstr(pSeq[a.revId]);
      o += ",\"RevTm\":" + dtoa::dump_double(a.revTemp) + ",\"Seq\":" + jstr(seqstr) + "}";
    }
    o += "]}";
  }
  o += "}\n";
  return o;
}

void emit_search(const SearchConfig& c, const std::string& json) {
  if (c.has_outfile) {  // silica.h:192-198: one gzip stream, file overwritten
    std::ofstream trunc(c.outfile.c_str(), std::ios_base::out | std::ios_base::binary | std::ios_base::trunc);
    trunc.close();
    append_gzip_member(c.outfile, json);
  } else {
    std::fwrite(json.data(), 1, json.size(), stdout);
    std::fflush(stdout);
  }
}

std::string revcomp_upper(std::string s) {  // util.h:110-114
  for (auto& ch : s) {
    char u = (char)std::toupper((unsigned char)ch);
    ch = u == 'A' ? 'T' : u == 'C' ? 'G' : u == 'G' ? 'C' : u == 'T' ? 'A' : u == 'U' ? 'A' : u == 'R' ? 'Y' : u == 'Y' ? 'R' : u == 'S' ? 'S'
         : u == 'W' ? 'W' : u == 'K' ? 'M' : u == 'M' ? 'K' : u == 'B' ? 'V' : u == 'V' ? 'B' : u == 'D' ? 'H' : u == 'H' ? 'D' : 'N';
  }
  return std::string(s.rbegin(), s.rend());
}

int silica(int argc, char** argv) {
  SearchConfig c;
  const OptSpec specs[] = {{"help", '?', false}, {"genome", 'g', true}, {"config", 'i', true}, {"outfile", 'o', true}, {"kmer", 'k', true},
                           {"maxmatches", 'm', true}, {"maxNeighborhood", 'x', true}, {"distance", 'd', true}, {"pruneprimer", 'q', true},
                           {"hamming", 'n', false}, {"cutTemp", 'c', true}, {"maxProdSize", 'l', true}, {"cutoffPenalty", 0, true},
                           {"penaltyTmDiff", 0, true}, {"penaltyTmMismatch", 0, true}, {"penaltyLength", 0, true}, {"enttemp", 0, true},
                           {"monovalent", 0, true}, {"divalent", 0, true}, {"dna", 0, true}, {"dntp", 0, true}, {"input-file", 0, true}};
  Parsed p = parse_options(argc, argv, specs, sizeof specs / sizeof specs[0]);
  if (!p.error.empty()) {
    std::cerr << "terminate called after throwing an instance of 'boost::program_options::error'\n  what():  " << p.error << std::endl;
    std::abort();
  }
  bool have_genome = false, have_input = false;
  for (auto& kv : p.kv) {
    const std::string& k = kv.first;
    const char* v = kv.second.c_str();
    if (k == "help") c.help = true;
    else if (k == "genome") { c.genome = v; have_genome = true; }
    else if (k == "config") c.primer3Config = v;
    else if (k == "outfile") { c.outfile = v; c.has_outfile = true; }
    else if (k == "kmer") c.kmer = (uint32_t)std::strtoul(v, nullptr, 10);
    else if (k == "maxmatches") c.max_locations = std::strtoull(v, nullptr, 10);
    else if (k == "maxNeighborhood") c.maxNeighborhood = (uint32_t)std::strtoul(v, nullptr, 10);
    else if (k == "distance") c.distance = (uint32_t)std::strtoul(v, nullptr, 10);
    else if (k == "pruneprimer") { c.maxPruneCount = (uint32_t)std::strtoul(v, nullptr, 10); c.pruneprimer = true; }
    else if (k == "hamming") c.hamming = true;
    else if (k == "cutTemp") c.cutTemp = std::strtod(v, nullptr);
    else if (k == "maxProdSize") c.maxProdSize = (uint32_t)std::strtoul(v, nullptr, 10);
    else if (k == "cutoffPenalty") c.cutofPen = std::strtod(v, nullptr);
    else if (k == "penaltyTmDiff") c.penDiff = std::strtod(v, nullptr);
    else if (k == "penaltyTmMismatch") c.penMis = std::strtod(v, nullptr);
    else if (k == "penaltyLength") c.penLen = std::strtod(v, nullptr);
    else if (k == "enttemp") c.temp = std::strtod(v, nullptr);
    else if (k == "monovalent") c.mv = std::strtod(v, nullptr);
    else if (k == "divalent") c.dv = std::strtod(v, nullptr);
    else if (k == "dna") c.dna_conc = std::strtod(v, nullptr);
    else if (k == "dntp") c.dntp = std::strtod(v, nullptr);
    else if (k == "input-file") { c.infile = v; have_input = true; }
  }
  if (!p.positional.empty()) {
    c.infile = p.positional.back();
    have_input = true;
  }
  if (c.help || !have_input || !have_genome) {
    std::cout << "Usage: dicey " << argv[0] << " [OPTIONS] -g <ref.fa.gz> sequences.fasta" << std::endl;
    std::cout << "  -g genome  -i primer3 config dir  -o outfile  -k kmer(15)  -m maxmatches(10000)  -x maxNeighborhood(10000)\n"
                 "  -d distance(1)  -q pruneprimer  -n hamming  -c cutTemp(45)  -l maxProdSize(15000)  --cutoffPenalty(-1)\n"
                 "  --penaltyTmDiff(0.6)  --penaltyTmMismatch(0.4)  --penaltyLength(0.001)  --enttemp(37)  --monovalent(50)\n"
                 "  --divalent(1.5)  --dna(50)  --dntp(0.6)\n\n";
    return -1;
  }
  std::vector<PrimerBind> allp;
  std::vector<PcrProduct> pcrColl;
  std::vector<std::string> msg, seqname, pName, pSeq;
  std::vector<uint32_t> seqlen;
  dg_index* ix = nullptr;
  auto bail = [&](const char* m) {
    msg.push_back(m);
    emit_search(c, search_json(c, ix, seqlen, seqname, allp, pcrColl, pName, pSeq, msg));
    if (ix) dg_index_close(ix);
    return 1;
  };
  if (!file_nonempty(c.genome)) return bail("Error: Genome does not exist!");
  {  // silica.h:303-315
    struct stat st;
    if (stat(c.primer3Config.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return bail("Error: Cannot find primer3 config directory!");
    while (c.primer3Config.size() > 1 && c.primer3Config.back() == '/') c.primer3Config.pop_back();
    c.primer3Config.push_back('/');
    if (!is_regular(c.primer3Config + "tetraloop.dh")) return bail("Error: Config directory path appears to be incorrect!");
  }
  if (!seq_len_name(c.genome, seqlen, seqname)) return bail("Error: Could not retrieve sequence lengths!");
  const int dev = device_from_env();
  if (dg_index_open((strip_last_extension(c.genome) + ".fm9").c_str(), dev, (std::getenv("DICEY_FULL_TABLE") ? DG_OPEN_BIG_TABLE : DG_OPEN_DEFAULT) | DG_OPEN_COMPACT, &ix) != DG_OK) {
    std::cerr << "dicey: " << dg_last_error() << std::endl;
    ix = nullptr;
    return bail("Error: FM-Index cannot be loaded!");
  }
  dg_thal* th = nullptr;
  if (dg_thal_open(c.primer3Config.c_str(), c.mv, c.dv, c.dntp, c.dna_conc, dev, &th) != DG_OK) {
    std::cerr << "dicey: " << dg_last_error() << std::endl;
    return bail("Error: Config directory path appears to be incorrect!");
  }
  {  // silica.h:350-353
    struct stat st;
    if (stat(c.infile.c_str(), &st) != 0 || S_ISDIR(st.st_mode)) {
      dg_thal_close(th);
      return bail("Error: Input fasta file is missing!");
    }
  }
  // ---- primer FASTA (silica.h:355-410), quirks included: a record counts only if longer than k, and the sequence
  // buffer is cleared only when a record is taken
  auto count_le = [&](const std::string& s, uint32_t limit) {
    uint64_t off[2] = {0, s.size()}, cnt = 0;
    if (dg_count(ix, (const uint8_t*)s.data(), off, 1, &cnt) != DG_OK) std::cerr << "dicey: " << dg_last_error() << std::endl;
    return cnt <= limit;
  };
  bool fatal = false;
  auto take = [&](const std::string& fan, const std::string& tmpfasta) -> bool {
    std::string qr = tmpfasta.substr(tmpfasta.size() - c.kmer);
    if (!c.pruneprimer || count_le(qr, c.maxPruneCount)) {
      qr = revcomp_upper(qr);
      if (!c.pruneprimer || count_le(qr, c.maxPruneCount)) {
        std::string inseq;
        for (char ch : tmpfasta) {  // util.h:208-219
          if (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') inseq.push_back(ch);
          else {
            msg.push_back("Warning: Non-DNA character in nucleotide sequence detected and replaced by 'N'!");
            inseq.push_back('N');
          }
        }
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
  {
    std::ifstream fa(c.infile.c_str());
    std::string fan, tmpfasta, line;
    while (!fatal && std::getline(fa, line)) {
      if (line.empty()) continue;
      if (line[0] == '>') {
        if (!fan.empty() && !tmpfasta.empty() && tmpfasta.size() > c.kmer) {
          if (!take(fan, tmpfasta)) fatal = true;
          tmpfasta = "";
        }
        fan = line.substr(1);
      } else {
        for (auto& ch : line) ch = (char)std::toupper((unsigned char)ch);
        tmpfasta += line;
      }
    }
    if (!fatal && !fan.empty() && !tmpfasta.empty() && tmpfasta.size() > c.kmer)
      if (!take(fan, tmpfasta)) fatal = true;
  }
  auto finish = [&](int rc) {
    emit_search(c, search_json(c, ix, seqlen, seqname, allp, pcrColl, pName, pSeq, msg));
    dg_thal_close(th);
    dg_index_close(ix);
    return rc;
  };
  if (fatal) return finish(1);
  const uint32_t nseq = (uint32_t)seqlen.size();
  std::vector<std::vector<PrimerBind>> forBind(nseq), revBind(nseq);
  if (!pSeq.empty()) {
    std::string pb;
    std::vector<uint64_t> poff(1, 0);
    for (auto& s : pSeq) {
      pb += s;
      poff.push_back(pb.size());
    }
    dg_search_params sp;
    sp.distance = c.distance;
    sp.hamming = c.hamming;
    sp.max_locations = c.max_locations;
    sp.max_neighborhood = c.maxNeighborhood;
    sp.kmer = c.kmer;
    sp.cut_temp = c.cutTemp;
    dg_search_result* R = nullptr;
    if (dg_search_sites(ix, th, &sp, seqlen.data(), nseq, (const uint8_t*)pb.data(), poff.data(), pSeq.size(), &R) != DG_OK) {
      std::cerr << "dicey: " << dg_last_error() << std::endl;
      dg_thal_close(th);
      dg_index_close(ix);
      return 2;
    }
    // per primer in order: a thal failure ends the run with the error JSON (silica.h:438-442, 512-516)
    uint64_t si = 0;
    for (size_t q = 0; q < pSeq.size(); ++q) {
      if (R->pflags[q] & DG_P_THAL_FAILED) {
        msg.push_back("Error: Thermodynamical calculation failed!");
        dg_search_result_free(R);
        return finish(1);
      }
      if (R->pflags[q] & DG_Q_NBHD_EXCEEDED) {  // silica.h:457-460
        std::string x = std::to_string(c.maxNeighborhood);
        msg.push_back("Warning: Neighborhood size exceeds " + x + " candidates. Only first " + x + " neighbors are searched, results are likely incomplete!");
      }
      for (; si < R->nsites && R->sites[si].primer == q; ++si) {
        const dg_site& s = R->sites[si];
        PrimerBind b;
        b.refIndex = s.ref;
        b.pos = s.pos;
        b.primerId = s.primer;
        b.onFor = s.on_for != 0;
        b.temp = s.temp;
        b.perfTemp = s.perf_temp;
        b.genome.assign(R->genome_pool + s.genome_off, s.genome_len);
        (b.onFor ? forBind : revBind)[s.ref].push_back(b);
      }
      if (R->pflags[q] & DG_Q_MAX_MATCHES) {
        std::string x = std::to_string(c.max_locations);
        msg.push_back("Warning: More than " + x + " matches found. Only first " + x + " matches are reported, results are likely incomplete!");
      }
    }
    dg_search_result_free(R);
  }
  for (uint32_t r = 0; r < nseq; ++r) {  // silica.h:581-584
    allp.insert(allp.end(), forBind[r].begin(), forBind[r].end());
    allp.insert(allp.end(), revBind[r].begin(), revBind[r].end());
  }
  std::sort(allp.begin(), allp.end());  // silica.h:587
  if (!c.pruneprimer) {
    // Amplicons (what silica.h:590-634 produces): per chromosome, every forward site in the order it was found, paired with
    // the reverse sites — taken in the order THEY were found — that start right of it and end within maxProdSize of it.
    // The reverse sites are looked up through a position-ordered view so that only the ones inside the window are touched;
    // the window's members are then put back into discovery order, which is the order the products must have before the
    // final (unstable) sort by penalty.
    struct Placed {
      uint32_t pos, found;  // position, discovery index within the chromosome's reverse sites
    };
    // the penalty of one primer pair: both Tm shortfalls against the perfect match, the Tm difference, the product length
    // (operands and order as in silica.h:623-629: the doubles are printed with 17 significant digits)
    auto pair_penalty = [&c](const PrimerBind& f, const PrimerBind& v, uint32_t span) {
      const double short_f = (f.perfTemp - f.temp) * c.penDiff, short_v = (v.perfTemp - v.temp) * c.penDiff;
      double total = short_f < 0 ? 0 : short_f;
      if (short_v > 0) total += short_v;
      total += std::abs(f.temp - v.temp) * c.penMis;
      total += span * c.penLen;
      return total;
    };
    std::vector<Placed> view;
    std::vector<uint32_t> window;
    for (uint32_t chrom = 0; chrom < nseq; ++chrom) {
      const std::vector<PrimerBind>& fwd = forBind[chrom];
      const std::vector<PrimerBind>& rev = revBind[chrom];
      view.resize(rev.size());
      for (uint32_t k = 0; k < rev.size(); ++k) view[k] = Placed{rev[k].pos, k};
      std::sort(view.begin(), view.end(), [](const Placed& x, const Placed& y) { return x.pos != y.pos ? x.pos < y.pos : x.found < y.found; });
      for (const PrimerBind& f : fwd) {
        const uint64_t reach = (uint64_t)f.pos + c.maxProdSize;  // a reverse site further right cannot end inside the limit
        const auto from = std::partition_point(view.begin(), view.end(), [&](const Placed& x) { return x.pos <= f.pos; });
        const auto upto = std::partition_point(from, view.end(), [&](const Placed& x) { return (uint64_t)x.pos <= reach; });
        window.clear();
        for (auto it = from; it != upto; ++it) window.push_back(it->found);
        std::sort(window.begin(), window.end());
        for (uint32_t k : window) {
          const PrimerBind& v = rev[k];
          const uint64_t span = (uint64_t)v.pos + pSeq[v.primerId].size() - f.pos;
          if (span > c.maxProdSize) continue;
          PcrProduct amp;
          amp.refIndex = chrom;
          amp.leng = (uint32_t)span;
          amp.forPos = f.pos;
          amp.revPos = v.pos;
          amp.forId = f.primerId;
          amp.revId = v.primerId;
          amp.forTemp = f.temp;
          amp.revTemp = v.temp;
          amp.penalty = pair_penalty(f, v, amp.leng);
          if (c.cutofPen < 0 || amp.penalty < c.cutofPen) pcrColl.push_back(amp);
        }
      }
    }
    std::sort(pcrColl.begin(), pcrColl.end());  // silica.h:637
  }
  return finish(0);
}

// ------------------------------------------------------------------------------------------------ index (index.h:34-141)
// `dicey index --verify genome.fa.gz`: the acceptance procedure for an index file this build did not write.  The
// reader's layout is restated from sdsl-lite: the first genuine file decides whether it is right, and this is
// what decides it in minutes — or names the section that is off.  1. dg_fm9_check, deep (host): byte accounting, rank words, tree,
// C[], samples.  2. dg_index_open: the device derives BWT / text / suffix array and checks them against the file (C[] against
// symbol totals, the file's SA samples, sampled suffix order).  3. the WHOLE text (dg_extract) against the text index.h:97-115
// builds from the FASTA.  4. 1 000 sampled 24-mers: dg_count, and every dg_locate position spells the pattern.
int verify_index(const std::string& fm9, const std::string& text) {
  std::vector<char> rep(1 << 16);
  std::cerr << "[1/4] sections of " << fm9 << " (host)" << std::endl;
  if (dg_fm9_check(fm9.c_str(), DG_FM9_CHECK_DEEP, rep.data(), rep.size()) != DG_OK) {
    std::cerr << "REFUSED: " << dg_last_error() << std::endl;
    std::cout << rep.data() << std::endl;
    return 1;
  }
  std::cerr << "[2/4] open on the device (derived layouts checked against the file)" << std::endl;
  dg_index* ix = nullptr;
  if (dg_index_open(fm9.c_str(), device_from_env(), 0, &ix) != DG_OK) {
    std::cerr << "REFUSED: " << dg_last_error() << std::endl;
    return 1;
  }
  dg_index_stats_t st;
  dg_index_stats(ix, &st);
  auto refuse = [&](const std::string& why) {
    std::cerr << "REFUSED: " << why << std::endl;
    dg_index_close(ix);
    return 1;
  };
  if (st.n != text.size() + 1) return refuse("the index holds " + std::to_string(st.n - 1) + " symbols, the FASTA gives " + std::to_string(text.size()));
  std::cerr << "[3/4] the whole text against the FASTA (" << text.size() << " symbols)" << std::endl;
  const uint64_t CH = 64ull << 20;
  std::vector<uint8_t> buf(std::min<uint64_t>(CH, text.size()));
  for (uint64_t at = 0; at < text.size(); at += CH) {
    const uint64_t lo = at, end = std::min<uint64_t>(text.size(), at + CH), hi = end - 1 /* inclusive */, off[1] = {0};
    if (dg_extract(ix, &lo, &hi, 1, buf.data(), off) != DG_OK) return refuse(dg_last_error());
    if (std::memcmp(buf.data(), text.data() + lo, end - lo) != 0) {
      uint64_t k = 0;
      while (buf[k] == (uint8_t)text[lo + k]) ++k;
      return refuse("text position " + std::to_string(lo + k) + ": the index spells byte " + std::to_string((unsigned)buf[k]) + ", the FASTA " +
                    std::to_string((unsigned)(uint8_t)text[lo + k]));
    }
  }
  std::cerr << "[4/4] 1000 sampled 24-mers: count and locate" << std::endl;
  const uint32_t m = 24;
  if (text.size() > m + 1) {
    std::string pat;
    std::vector<uint64_t> poff(1, 0), where;
    uint64_t x = 0x9E3779B97F4A7C15ull;
    while (where.size() < 1000) {
      x ^= x << 13, x ^= x >> 7, x ^= x << 17;
      const uint64_t p0 = x % (text.size() - m);
      pat.append(text, p0, m);
      poff.push_back(pat.size());
      where.push_back(p0);
    }
    std::vector<uint64_t> cnt(where.size());
    if (dg_count(ix, (const uint8_t*)pat.data(), poff.data(), where.size(), cnt.data()) != DG_OK) return refuse(dg_last_error());
    dg_locations* loc = nullptr;
    if (dg_locate(ix, (const uint8_t*)pat.data(), poff.data(), where.size(), &loc) != DG_OK) return refuse(dg_last_error());
    std::string why;
    for (size_t i = 0; i < where.size() && why.empty(); ++i) {
      const uint64_t a = loc->off[i], b = loc->off[i + 1];
      bool own = false;
      if (b - a != cnt[i]) why = "pattern " + std::to_string(i) + ": count says " + std::to_string(cnt[i]) + ", locate returns " + std::to_string(b - a);
      for (uint64_t k = a; k < b && why.empty(); ++k) {
        own = own || loc->pos[k] == where[i];
        if (loc->pos[k] + m > text.size() || text.compare(loc->pos[k], m, pat, poff[i], m) != 0)
          why = "pattern " + std::to_string(i) + ": located position " + std::to_string(loc->pos[k]) + " does not spell it";
      }
      if (why.empty() && !own) why = "pattern " + std::to_string(i) + " was cut from text position " + std::to_string(where[i]) + ", which locate does not return";
    }
    dg_locations_free(loc);
    if (!why.empty()) return refuse(why);
  }
  dg_index_close(ix);
  std::cout << "Verified: " << fm9 << " is the index of this FASTA (" << text.size() << " symbols)." << std::endl;
  return 0;
}

int indexer(int argc, char** argv) {
  std::string genome, outfile;
  bool out_given = false, help = false;
  bool verify = false;
  const OptSpec specs[] = {{"help", '?', false}, {"output", 'o', true}, {"input-file", 0, true}, {"verify", 'v', false}};
  Parsed p = parse_options(argc, argv, specs, 4);
  if (!p.error.empty()) {
    std::cerr << p.error << std::endl;
    std::abort();
  }
  for (auto& kv : p.kv) {
    if (kv.first == "help") help = true;
    else if (kv.first == "output") { outfile = kv.second; out_given = true; }
    else if (kv.first == "input-file") genome = kv.second;
    else if (kv.first == "verify") verify = true;
  }
  if (!p.positional.empty()) genome = p.positional.back();
  if (help || genome.empty()) {
    std::cout << "Usage: dicey " << argv[0] << " [OPTIONS] genome.fa.gz" << std::endl;
    std::cout << "Generic options:\n  -? [ --help ]                    show help message\n  -o [ --output ] arg (=genome.fm9) output file\n"
                 "  -v [ --verify ]                  do not build: accept or refuse the EXISTING index file against this FASTA\n"
                 "                                   (every section of the sdsl file, then the whole text and sampled\n"
                 "                                   count / locate answers on the GPU) - the check for the first\n"
                 "                                   index that `dicey index` of the reference wrote\n\n";
    return -1;
  }
  if (!out_given) outfile = strip_last_extension(genome) + ".fm9";  // index.h:67-69
  std::ifstream probe(genome.c_str(), std::ios::binary);
  if (!probe.is_open()) {
    std::cerr << "Error: " << genome << " cannot be opened!" << std::endl;
    return -1;
  }
  probe.close();
  if (!is_gz(genome)) {
    std::cerr << "Error: Please compress " << genome << " with bgzip." << std::endl;
    return -1;
  }
  // index.h:97-115: header lines become '\n' (except before the first sequence), sequence lines are upper-cased
  std::string text, line;
  LineReader r(genome);
  bool first = true;
  while (r.next(line)) {
    if (!line.empty() && line[0] == '>') {
      if (!first) text.push_back('\n');
      else first = false;
    } else {
      for (char ch : line) text.push_back((char)std::toupper((unsigned char)ch));
    }
  }
  text.push_back('\n');
  if (verify) return verify_index(outfile, text);
  if (dg_index_build((const uint8_t*)text.data(), text.size(), device_from_env(), outfile.c_str()) != DG_OK) {
    std::cerr << "dicey: " << dg_last_error() << std::endl;
    return 1;
  }
  std::cout << "Done." << std::endl;
  return 0;
}

void display_usage() {  // dicey.cpp:19-33 (only the subcommands this build carries)
  std::cout << "Usage: dicey <command> <arguments>" << std::endl;
  std::cout << std::endl;
  std::cout << "    index        index FASTA reference file (GPU builder)" << std::endl;
  std::cout << "    hunt         search DNA sequences (MI355X search path)" << std::endl;
  std::cout << "    search       in-silico PCR (MI355X search path)" << std::endl;
  std::cout << "    padlock      padlock probe design (MI355X search path)" << std::endl;
  std::cout << std::endl;
  std::cout << "chop and mappability are not part of this build; use the reference binary for them." << std::endl;
  std::cout << std::endl;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    display_usage();
    return 0;
  }
  std::string cmd = argv[1];
  if (cmd == "version" || cmd == "--version" || cmd == "-v") {
    std::cout << "Dicey version: v" << kVersion << " (MI355X search path, libdiceygpu ABI " << dg_abi_version() << ")" << std::endl;
    return 0;
  }
  if (cmd == "help" || cmd == "--help" || cmd == "-h" || cmd == "-?") {
    display_usage();
    return 0;
  }
  if (cmd == "hunt") {
    const double t_main0 = now_ms();
    const int rc = hunter(argc - 1, argv + 1);
    if (timing_on()) std::fprintf(stderr, "dicey timing: %-28s %8.1f ms\n", "hunt, entry to return", now_ms() - t_main0);
    // every line is written and every file closed: leave without the runtime's teardown (unloading the code objects, releasing
    // 98 GB allocation by allocation — 0.1-0.2 s of a 10 M-query run; the driver releases the process's memory either way)
    if (quick_exit_on()) {
      std::cout.flush();
      std::fflush(stdout);
      std::fflush(stderr);
      _exit(rc);
    }
    return rc;
  }
  if (cmd == "index") return indexer(argc - 1, argv + 1);
  if (cmd == "search") return silica(argc - 1, argv + 1);
  if (cmd == "padlock") return padlock_main(argc - 1, argv + 1);
  std::cerr << "Unrecognized command " << cmd << std::endl;
  return 1;
}
