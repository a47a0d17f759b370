// Helpers shared by the subcommands of the `dicey` host binary (dicey_main.cpp: hunt, search, index; padlock.cpp).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <vector>

namespace {

const char* kVersion = "0.5.1";  // src/version.h:8 — goes into meta.version

// ------------------------------------------------------------------------------------------------ small utilities
inline bool file_nonempty(const std::string& p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0;
}
inline bool is_regular(const std::string& p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
// boost::filesystem parent_path()/stem() of the genome path (hunter.h:254-255): "dir/hg19.fa.gz" -> "dir/hg19.fa"
inline std::string strip_last_extension(const std::string& p) {
  size_t slash = p.find_last_of('/');
  size_t start = slash == std::string::npos ? 0 : slash + 1;
  std::string name = p.substr(start);
  size_t dot = name.find_last_of('.');
  if (dot != std::string::npos && dot != 0 && name != "..") name = name.substr(0, dot);
  return p.substr(0, start) + name;
}
inline bool is_gz(const std::string& p) {  // util.h:21-33
  std::ifstream f(p.c_str(), std::ios::binary);
  char a = 0, b = 0;
  f.read(&a, 1);
  f.read(&b, 1);
  return a == '\x1F' && b == '\x8B';
}
// reads a (b)gzip or plain file line by line
struct LineReader {
  gzFile g = nullptr;
  std::string buf;
  explicit LineReader(const std::string& p) { g = gzopen(p.c_str(), "rb"); if (g) gzbuffer(g, 1 << 20); }
  ~LineReader() { if (g) gzclose(g); }
  bool ok() const { return g != nullptr; }
  bool next(std::string& line) {
    line.clear();
    char tmp[1 << 16];
    bool any = false;
    while (gzgets(g, tmp, sizeof tmp)) {
      any = true;
      size_t n = std::strlen(tmp);
      if (n && tmp[n - 1] == '\n') {
        line.append(tmp, n - 1);
        return true;
      }
      line.append(tmp, n);
    }
    return any;
  }
};
inline bool is_fasta(const std::string& p) {  // util.h:35-52: first line starts with '>'
  LineReader r(p);
  std::string line;
  if (!r.ok() || !r.next(line)) return false;
  return !line.empty() && line[0] == '>';
}

// sequence names and lengths as htslib's faidx reports them (util.h:183-206): from <genome>.fai when present,
// otherwise by scanning the FASTA.  Name = header up to the first whitespace.
inline bool seq_len_name(const std::string& genome, std::vector<uint32_t>& seqlen, std::vector<std::string>& seqname) {
  std::ifstream fai((genome + ".fai").c_str());
  if (fai) {
    std::string line;
    while (std::getline(fai, line)) {
      if (line.empty()) continue;
      std::istringstream ss(line);
      std::string name;
      unsigned long long len = 0;
      if (!std::getline(ss, name, '\t') || !(ss >> len)) return false;
      seqname.push_back(name);
      seqlen.push_back((uint32_t)len + 1);  // util.h:201
    }
    return !seqlen.empty();
  }
  LineReader r(genome);
  if (!r.ok()) return false;
  std::string line;
  bool have = false;
  uint64_t len = 0;
  while (r.next(line)) {
    if (!line.empty() && line[0] == '>') {
      if (have) seqlen.push_back((uint32_t)len + 1);
      size_t e = line.find_first_of(" \t\r", 1);
      seqname.push_back(line.substr(1, e == std::string::npos ? std::string::npos : e - 1));
      have = true;
      len = 0;
    } else if (have) {
      for (char c : line) len += !(c == '\r' || c == ' ' || c == '\t');
    }
  }
  if (have) seqlen.push_back((uint32_t)len + 1);
  return !seqlen.empty();
}

// ------------------------------------------------------------------------------------------------ JSON (nlohmann 3.5.0 dump())
inline void jstr_append(std::string& o, const char* p, size_t n) {
  static const char* hex = "0123456789abcdef";
  o.push_back('"');
  for (size_t i = 0; i < n; ++i) {
    const unsigned char c = (unsigned char)p[i];
    if (c == '"') o += "\\\"";
    else if (c == '\\') o += "\\\\";
    else if (c == '\b') o += "\\b";
    else if (c == '\f') o += "\\f";
    else if (c == '\n') o += "\\n";
    else if (c == '\r') o += "\\r";
    else if (c == '\t') o += "\\t";
    else if (c < 0x20) {
      o += "\\u00";
      o.push_back(hex[c >> 4]);
      o.push_back(hex[c & 15]);
    } else o.push_back((char)c);
  }
  o.push_back('"');
}
inline void jstr_append(std::string& o, const std::string& s) { jstr_append(o, s.data(), s.size()); }
inline std::string jstr(const std::string& s) {
  std::string o;
  jstr_append(o, s);
  return o;
}
inline void uint_append(std::string& o, uint64_t v) {  // std::to_string without the temporary
  char buf[24];
  int n = 0;
  do {
    buf[n++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) o.push_back(buf[--n]);
}

// one gzip member per call, appended (hunter.h:162-170: gzip_compressor + file_sink(app))
inline bool append_gzip_member(const std::string& path, const std::string& data) {
  FILE* f = std::fopen(path.c_str(), "ab");
  if (!f) return false;
  static const unsigned char hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
  std::fwrite(hdr, 1, 10, f);
  z_stream zs;
  std::memset(&zs, 0, sizeof zs);
  deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
  std::vector<unsigned char> out(deflateBound(&zs, data.size()) + 64);
  zs.next_in = (Bytef*)data.data();
  zs.avail_in = (uInt)data.size();
  zs.next_out = out.data();
  zs.avail_out = (uInt)out.size();
  deflate(&zs, Z_FINISH);
  std::fwrite(out.data(), 1, zs.total_out, f);
  deflateEnd(&zs);
  uint32_t crc = (uint32_t)crc32(0L, (const Bytef*)data.data(), (uInt)data.size()), isz = (uint32_t)data.size();
  std::fwrite(&crc, 4, 1, f);
  std::fwrite(&isz, 4, 1, f);
  return std::fclose(f) == 0;
}

// ------------------------------------------------------------------------------------------------ options (boost::program_options subset)
struct OptSpec {
  const char* lname;
  char sname;
  bool takes_arg;
};
struct Parsed {
  std::vector<std::pair<std::string, std::string>> kv;
  std::vector<std::string> positional;
  std::string error;
};
inline Parsed parse_options(int argc, char** argv, const OptSpec* specs, size_t nspec) {
  Parsed p;
  auto find_long = [&](const std::string& n) -> const OptSpec* {
    const OptSpec* exact = nullptr;
    const OptSpec* pref = nullptr;
    int npref = 0;
    for (size_t i = 0; i < nspec; ++i) {
      if (n == specs[i].lname) exact = &specs[i];
      else if (std::string(specs[i].lname).compare(0, n.size(), n) == 0) {
        pref = &specs[i];
        ++npref;
      }
    }
    if (exact) return exact;
    return npref == 1 ? pref : nullptr;  // unambiguous prefixes are accepted, as program_options does
  };
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
      std::string name = a.substr(2), val;
      bool has_val = false;
      size_t eq = name.find('=');
      if (eq != std::string::npos) {
        val = name.substr(eq + 1);
        name = name.substr(0, eq);
        has_val = true;
      }
      const OptSpec* s = find_long(name);
      if (!s) {
        p.error = "unrecognised option '" + a + "'";
        return p;
      }
      if (s->takes_arg && !has_val) {
        if (i + 1 >= argc) {
          p.error = std::string("the required argument for option '--") + s->lname + "' is missing";
          return p;
        }
        val = argv[++i];
      }
      p.kv.emplace_back(s->lname, val);
    } else if (a.size() >= 2 && a[0] == '-' && a != "--") {
      size_t k = 1;
      while (k < a.size()) {
        const OptSpec* s = nullptr;
        for (size_t j = 0; j < nspec; ++j)
          if (specs[j].sname && specs[j].sname == a[k]) s = &specs[j];
        if (!s) {
          p.error = "unrecognised option '" + a + "'";
          return p;
        }
        if (s->takes_arg) {
          std::string val = a.substr(k + 1);
          if (val.empty()) {
            if (i + 1 >= argc) {
              p.error = std::string("the required argument for option '--") + s->lname + "' is missing";
              return p;
            }
            val = argv[++i];
          }
          p.kv.emplace_back(s->lname, val);
          break;
        }
        p.kv.emplace_back(s->lname, "");
        ++k;
      }
    } else if (a == "--") {
      for (++i; i < argc; ++i) p.positional.push_back(argv[i]);
    } else p.positional.push_back(a);
  }
  return p;
}


}  // namespace
