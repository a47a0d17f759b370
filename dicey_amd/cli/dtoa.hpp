// Shortest-ish decimal representation of a double the way nlohmann::json 3.5.0's dump() prints it (the reference writes
// every Tm / penalty through json::dump(), src/silica.h:143,149,160-170): Grisu2 (Loitsch 2010) digit generation with
// the customary ±1 ulp safety margin and round-weed step, then fixed notation for decimal exponents in (-4, 15] and
// exponent notation otherwise.  Written from the published algorithm; tests/test_dtoa.py checks it against the
// reference's own vendored header on random bit patterns.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

namespace dtoa {

struct DiyFp {
  uint64_t f;
  int e;
};
struct CachedPower {
  uint64_t f;
  int e;
  int k;
};
// 10^k for k = -300, -292, ..., 324 as normalised 64-bit significands (round to nearest)
static const CachedPower kPow10[] = {
    {0xAB70FE17C79AC6CAULL, -1060, -300},
    {0xFF77B1FCBEBCDC4FULL, -1034, -292},
    {0xBE5691EF416BD60CULL, -1007, -284},
    {0x8DD01FAD907FFC3CULL, -980, -276},
    {0xD3515C2831559A83ULL, -954, -268},
    {0x9D71AC8FADA6C9B5ULL, -927, -260},
    {0xEA9C227723EE8BCBULL, -901, -252},
    {0xAECC49914078536DULL, -874, -244},
    {0x823C12795DB6CE57ULL, -847, -236},
    {0xC21094364DFB5637ULL, -821, -228},
    {0x9096EA6F3848984FULL, -794, -220},
    {0xD77485CB25823AC7ULL, -768, -212},
    {0xA086CFCD97BF97F4ULL, -741, -204},
    {0xEF340A98172AACE5ULL, -715, -196},
    {0xB23867FB2A35B28EULL, -688, -188},
    {0x84C8D4DFD2C63F3BULL, -661, -180},
    {0xC5DD44271AD3CDBAULL, -635, -172},
    {0x936B9FCEBB25C996ULL, -608, -164},
    {0xDBAC6C247D62A584ULL, -582, -156},
    {0xA3AB66580D5FDAF6ULL, -555, -148},
    {0xF3E2F893DEC3F126ULL, -529, -140},
    {0xB5B5ADA8AAFF80B8ULL, -502, -132},
    {0x87625F056C7C4A8BULL, -475, -124},
    {0xC9BCFF6034C13053ULL, -449, -116},
    {0x964E858C91BA2655ULL, -422, -108},
    {0xDFF9772470297EBDULL, -396, -100},
    {0xA6DFBD9FB8E5B88FULL, -369, -92},
    {0xF8A95FCF88747D94ULL, -343, -84},
    {0xB94470938FA89BCFULL, -316, -76},
    {0x8A08F0F8BF0F156BULL, -289, -68},
    {0xCDB02555653131B6ULL, -263, -60},
    {0x993FE2C6D07B7FACULL, -236, -52},
    {0xE45C10C42A2B3B06ULL, -210, -44},
    {0xAA242499697392D3ULL, -183, -36},
    {0xFD87B5F28300CA0EULL, -157, -28},
    {0xBCE5086492111AEBULL, -130, -20},
    {0x8CBCCC096F5088CCULL, -103, -12},
    {0xD1B71758E219652CULL, -77, -4},
    {0x9C40000000000000ULL, -50, 4},
    {0xE8D4A51000000000ULL, -24, 12},
    {0xAD78EBC5AC620000ULL, 3, 20},
    {0x813F3978F8940984ULL, 30, 28},
    {0xC097CE7BC90715B3ULL, 56, 36},
    {0x8F7E32CE7BEA5C70ULL, 83, 44},
    {0xD5D238A4ABE98068ULL, 109, 52},
    {0x9F4F2726179A2245ULL, 136, 60},
    {0xED63A231D4C4FB27ULL, 162, 68},
    {0xB0DE65388CC8ADA8ULL, 189, 76},
    {0x83C7088E1AAB65DBULL, 216, 84},
    {0xC45D1DF942711D9AULL, 242, 92},
    {0x924D692CA61BE758ULL, 269, 100},
    {0xDA01EE641A708DEAULL, 295, 108},
    {0xA26DA3999AEF774AULL, 322, 116},
    {0xF209787BB47D6B85ULL, 348, 124},
    {0xB454E4A179DD1877ULL, 375, 132},
    {0x865B86925B9BC5C2ULL, 402, 140},
    {0xC83553C5C8965D3DULL, 428, 148},
    {0x952AB45CFA97A0B3ULL, 455, 156},
    {0xDE469FBD99A05FE3ULL, 481, 164},
    {0xA59BC234DB398C25ULL, 508, 172},
    {0xF6C69A72A3989F5CULL, 534, 180},
    {0xB7DCBF5354E9BECEULL, 561, 188},
    {0x88FCF317F22241E2ULL, 588, 196},
    {0xCC20CE9BD35C78A5ULL, 614, 204},
    {0x98165AF37B2153DFULL, 641, 212},
    {0xE2A0B5DC971F303AULL, 667, 220},
    {0xA8D9D1535CE3B396ULL, 694, 228},
    {0xFB9B7CD9A4A7443CULL, 720, 236},
    {0xBB764C4CA7A44410ULL, 747, 244},
    {0x8BAB8EEFB6409C1AULL, 774, 252},
    {0xD01FEF10A657842CULL, 800, 260},
    {0x9B10A4E5E9913129ULL, 827, 268},
    {0xE7109BFBA19C0C9DULL, 853, 276},
    {0xAC2820D9623BF429ULL, 880, 284},
    {0x80444B5E7AA7CF85ULL, 907, 292},
    {0xBF21E44003ACDD2DULL, 933, 300},
    {0x8E679C2F5E44FF8FULL, 960, 308},
    {0xD433179D9C8CB841ULL, 986, 316},
    {0x9E19DB92B4E31BA9ULL, 1013, 324}
};

inline DiyFp mul(DiyFp x, DiyFp y) {
  const uint64_t M = 0xFFFFFFFFu;
  uint64_t a = x.f >> 32, b = x.f & M, c = y.f >> 32, d = y.f & M;
  uint64_t ac = a * c, bc = b * c, ad = a * d, bd = b * d;
  uint64_t mid = (bd >> 32) + (ad & M) + (bc & M);
  mid += 1ULL << 31;  // round the discarded low half
  return DiyFp{ac + (ad >> 32) + (bc >> 32) + (mid >> 32), x.e + y.e + 64};
}
inline DiyFp normalize(DiyFp x) {
  while ((x.f >> 63) == 0) {
    x.f <<= 1;
    --x.e;
  }
  return x;
}
inline DiyFp normalize_to(DiyFp x, int e) { return DiyFp{x.f << (x.e - e), e}; }

inline CachedPower cached_power_for(int e) {
  // smallest k with alpha <= e + e_k + 64, alpha = -60: k = ceil((-61 - e) * log10(2)), log10(2) ~ 78913 / 2^18
  const int f = -60 - e - 1;
  const int k = (f * 78913) / (1 << 18) + (f > 0 ? 1 : 0);
  const int idx = (300 + k + 7) / 8;
  return kPow10[idx];
}

inline void round_weed(char* buf, int len, uint64_t dist, uint64_t delta, uint64_t rest, uint64_t ten_k) {
  while (rest < dist && delta - rest >= ten_k && (rest + ten_k < dist || dist - rest > rest + ten_k - dist)) {
    --buf[len - 1];
    rest += ten_k;
  }
}

// digits of a positive finite double: buf[0..len) and the decimal exponent of the last digit
inline void grisu2(double value, char* buf, int& len, int& dec_exp) {
  uint64_t bits;
  std::memcpy(&bits, &value, 8);
  const uint64_t F = bits & ((1ULL << 52) - 1);
  const int E = (int)(bits >> 52) & 0x7FF;
  const bool denormal = E == 0;
  DiyFp v = denormal ? DiyFp{F, 1 - 1075} : DiyFp{F + (1ULL << 52), E - 1075};
  const bool lower_closer = (F == 0 && E > 1);
  DiyFp m_plus{2 * v.f + 1, v.e - 1};
  DiyFp m_minus = lower_closer ? DiyFp{4 * v.f - 1, v.e - 2} : DiyFp{2 * v.f - 1, v.e - 1};
  DiyFp w_plus = normalize(m_plus);
  DiyFp w_minus = normalize_to(m_minus, w_plus.e);
  DiyFp w = normalize(v);
  const CachedPower c = cached_power_for(w_plus.e);
  const DiyFp ck{c.f, c.e};
  DiyFp W = mul(w, ck), Wm = mul(w_minus, ck), Wp = mul(w_plus, ck);
  DiyFp Mm{Wm.f + 1, Wm.e}, Mp{Wp.f - 1, Wp.e};
  dec_exp = -c.k;
  // digit generation on M+ with the unsafe interval [M-, M+]
  uint64_t delta = Mp.f - Mm.f, dist = Mp.f - W.f;
  const DiyFp one{1ULL << -Mp.e, Mp.e};
  uint32_t p1 = (uint32_t)(Mp.f >> -one.e);
  uint64_t p2 = Mp.f & (one.f - 1);
  uint32_t pow10 = 1;
  int n = 1;
  while (n < 10 && p1 >= pow10 * 10ULL) {
    pow10 *= 10;
    ++n;
  }
  len = 0;
  while (n > 0) {
    const uint32_t d = p1 / pow10;
    p1 %= pow10;
    buf[len++] = (char)('0' + d);
    --n;
    const uint64_t rest = ((uint64_t)p1 << -one.e) + p2;
    if (rest <= delta) {
      dec_exp += n;
      round_weed(buf, len, dist, delta, rest, (uint64_t)pow10 << -one.e);
      return;
    }
    pow10 /= 10;
  }
  int m = 0;
  for (;;) {
    p2 *= 10;
    const uint64_t d = p2 >> -one.e;
    p2 &= one.f - 1;
    buf[len++] = (char)('0' + d);
    ++m;
    delta *= 10;
    dist *= 10;
    if (p2 <= delta) break;
  }
  dec_exp -= m;
  round_weed(buf, len, dist, delta, p2, one.f);
}

// json::dump() of a double (finite values; NaN/inf print "null" in nlohmann)
inline std::string dump_double(double x) {
  if (!(x == x) || x - x != 0) return "null";
  std::string out;
  if (x == 0) {
    uint64_t b;
    std::memcpy(&b, &x, 8);
    return (b >> 63) ? "-0.0" : "0.0";
  }
  if (x < 0) {
    out.push_back('-');
    x = -x;
  }
  char d[32];
  int k = 0, e = 0;
  grisu2(x, d, k, e);
  const int n = k + e;  // position of the decimal point relative to the first digit
  if (k <= n && n <= 15) {
    out.append(d, k);
    out.append((size_t)(n - k), '0');
    out += ".0";
  } else if (0 < n && n <= 15) {
    out.append(d, n);
    out.push_back('.');
    out.append(d + n, k - n);
  } else if (-4 < n && n <= 0) {
    out += "0.";
    out.append((size_t)(-n), '0');
    out.append(d, k);
  } else {
    out.push_back(d[0]);
    if (k > 1) {
      out.push_back('.');
      out.append(d + 1, k - 1);
    }
    out.push_back('e');
    int ex = n - 1;
    out.push_back(ex < 0 ? '-' : '+');
    if (ex < 0) ex = -ex;
    if (ex < 10) out.push_back('0');
    out += std::to_string(ex);
  }
  return out;
}

}  // namespace dtoa
