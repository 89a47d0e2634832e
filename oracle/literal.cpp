// CPU oracle (TEST INFRASTRUCTURE ONLY) -- literal, stage-by-stage restatement of
// muscato_window_reads -> muscato_screen -> sort -> muscato_confirm.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library; nothing under muscato_amd/ links or calls it.  It is the checker
// and the reported CPU baseline ("port"), never the product.
//
// Every stage follows the reference Go source (file:line cited per function,
// paths relative to the kshedden/muscato checkout).  Records are kept as
// (gene, pos) / read indices and compared exactly as GNU `sort` under LC_ALL=C
// would compare the text lines the reference writes, so that the order
// dependent MaxMatches truncation (cmd/muscato_confirm/main.go:233-242,
// 424-448) is reproduced.  Text (de)serialisation and snappy are NOT emulated:
// they are lossless containers, so this port does strictly less work than the Go
// pipeline (a conservative, i.e. fast, CPU baseline).
//
// Third-party algorithms absent from the reference tree (no go.mod, versions
// unpinned): github.com/chmduquesne/rollinghash/buzhash32
// (cyclic polynomial: sum = rotl(sum,1) ^ rotl(T[out], n%32) ^ T[in]) and
// golang-collections/go-datastructures/bitarray (dense bit array) are restated
// from their published algorithms.  Neither can change results (SURVEY.md 8a
// note H): the Bloom filter has no false negatives and every false positive is
// removed by the exact merge-join on the window key.
//
// Parity pinning: the reference's fixtures tests/data/muscato/00-04 are
// reproduced through this library by tests/test_oracle_golden.py; it is
// cross-checked against the independent Python direct formulation
// (oracle/muscato_oracle.py) on randomized inputs by tests/test_oracle_cross.py.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" {

#define ORC_MAX_WINDOWS 32

struct orc_params {
  int32_t n_windows;
  int32_t windows[ORC_MAX_WINDOWS];
  int32_t window_width;
  double pmatch;
  int32_t min_dinuc;
  int32_t max_read_length;
  int32_t max_matches;
  int32_t match_mode_first;  // 0 = "best", 1 = "first"
  uint64_t bloom_size;       // bits (cmd/muscato/main.go:859-862 default 4e9)
  int32_t num_hash;          // cmd/muscato/main.go:863-866 default 20
  int32_t nthreads;
};

struct orc_hit {
  uint32_t read_idx, gene_idx, pos, nmiss;
};

}  // extern "C"

namespace {

using clk = std::chrono::steady_clock;
static double secs(clk::time_point a, clk::time_point b) {
  return std::chrono::duration<double>(b - a).count();
}

struct Seqs {
  const char* buf;
  const uint64_t* off;
  uint64_t n;
  const char* ptr(uint64_t i) const { return buf + off[i]; }
  uint32_t len(uint64_t i) const { return (uint32_t)(off[i + 1] - off[i]); }
};

// utils/entropy.go:5-40
static int count_dinuc(const char* seq, int n) {
  int wk[25] = {0};
  int last = 0, cnt = 0;
  for (int i = 0; i < n; i++) {
    int v;
    switch (seq[i]) {
      case 'A': v = 0; break;
      case 'T': v = 1; break;
      case 'G': v = 2; break;
      case 'C': v = 3; break;
      default: v = 4;
    }
    if (i > 0) {
      int k = 5 * last + v;
      if (wk[k] == 0) cnt++;
      wk[k]++;
    }
    last = v;
  }
  return cnt;
}

// buzhash32 (see header).  State for one hash over a window of n bytes.
static inline uint32_t rotl32(uint32_t x, unsigned r) {
  r &= 31;
  return r ? (x << r) | (x >> (32 - r)) : x;
}

struct Tables {
  std::vector<uint32_t> t;  // num_hash * 256
  int nh;
  // cmd/muscato_screen/main.go:86-101: 256 distinct random uint32 per hash.
  // Go's unseeded math/rand stream is not reproducible here and is result
  // irrelevant (note H); splitmix64 with a fixed seed stands in.
  explicit Tables(int num_hash) : t((size_t)num_hash * 256), nh(num_hash) {
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
      uint64_t z = (s += 0x9E3779B97F4A7C15ull);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      return z ^ (z >> 31);
    };
    for (int j = 0; j < num_hash; j++) {
      std::vector<uint32_t> seen;
      for (int i = 0; i < 256; i++) {
        for (;;) {
          uint32_t x = (uint32_t)next();
          if (std::find(seen.begin(), seen.end(), x) == seen.end()) {
            seen.push_back(x);
            t[(size_t)j * 256 + i] = x;
            break;
          }
        }
      }
    }
  }
  const uint32_t* tab(int j) const { return &t[(size_t)j * 256]; }
};

// dense bit array (bitarray.NewBitArray / SetBit / GetBit)
struct BitArray {
  std::vector<uint64_t> w;
  uint64_t nbits;
  explicit BitArray(uint64_t n) : w((n + 63) / 64, 0), nbits(n) {}
  void set(uint64_t i) { w[i >> 6] |= 1ull << (i & 63); }
  bool get(uint64_t i) const { return (w[i >> 6] >> (i & 63)) & 1; }
};

struct Cand {  // one bmatch_k line: mseq, left, right, tnum, pos
  uint32_t gene, pos;
};

struct Ctx {
  Seqs reads, genes;
  const orc_params* P;
  int ww;
};

// right tail [jy, jz) of a candidate, exactly as processSeq stores it
// (cmd/muscato_screen/main.go:303-314 for pos 0, :347-353 otherwise).
static inline void cand_right(const Ctx& c, int k, const Cand& a, uint32_t* jy, uint32_t* jz) {
  const int q2 = c.P->windows[k] + c.ww;
  const uint32_t glen = c.genes.len(a.gene);
  if (a.pos == 0) {
    int64_t z = 100 - q2;  // the literal 100
    if (z > (int64_t)glen) z = glen;
    *jy = c.ww;
    *jz = (uint32_t)z;
  } else {
    int64_t y = (int64_t)a.pos + c.ww;
    int64_t z = y + c.P->max_read_length - q2;
    if (z > (int64_t)glen) z = glen;
    *jy = (uint32_t)y;
    *jz = (uint32_t)z;
  }
}

static inline int cmp_bytes(const char* a, size_t na, const char* b, size_t nb) {
  size_t n = na < nb ? na : nb;
  int r = n ? memcmp(a, b, n) : 0;
  if (r) return r;
  // the shorter field is followed by '\t' (or '\n'), which sorts below every base
  return na < nb ? -1 : (na > nb ? 1 : 0);
}

// LC_ALL=C whole-line order of "key\tleft\tright\t%011d\tpos"
// (cmd/muscato/main.go:341-350: sort -k1 == whole line).
static int cmp_cand(const Ctx& c, int k, const Cand& a, const Cand& b) {
  const int q1 = c.P->windows[k];
  const char* ga = c.genes.ptr(a.gene);
  const char* gb = c.genes.ptr(b.gene);
  int r = memcmp(ga + a.pos, gb + b.pos, c.ww);
  if (r) return r;
  // left: q1 bytes for the general path, empty for the pos-0 path (then q1 == 0)
  if (q1) {
    r = memcmp(ga + a.pos - q1, gb + b.pos - q1, q1);
    if (r) return r;
  }
  uint32_t ay, az, by, bz;
  cand_right(c, k, a, &ay, &az);
  cand_right(c, k, b, &by, &bz);
  r = cmp_bytes(ga + ay, az - ay, gb + by, bz - by);
  if (r) return r;
  if (a.gene != b.gene) return a.gene < b.gene ? -1 : 1;  // %011d: fixed width
  char sa[16], sb[16];
  int na = snprintf(sa, sizeof sa, "%u", a.pos), nb = snprintf(sb, sizeof sb, "%u", b.pos);
  return cmp_bytes(sa, na, sb, nb);
}

// LC_ALL=C whole-line order of "key\tleft\tright" (cmd/muscato/main.go:261-270)
static int cmp_src(const Ctx& c, int k, uint32_t a, uint32_t b) {
  const int q1 = c.P->windows[k], q2 = q1 + c.ww;
  const char* ra = c.reads.ptr(a);
  const char* rb = c.reads.ptr(b);
  int r = memcmp(ra + q1, rb + q1, c.ww);
  if (r) return r;
  if (q1) {
    r = memcmp(ra, rb, q1);
    if (r) return r;
  }
  return cmp_bytes(ra + q2, c.reads.len(a) - q2, rb + q2, c.reads.len(b) - q2);
}

template <class T, class Less>
static void parallel_sort(std::vector<T>& v, Less less, int nthreads) {
  size_t n = v.size();
  if (nthreads <= 1 || n < 100000) {
    std::sort(v.begin(), v.end(), less);
    return;
  }
  int parts = 1;
  while (parts < nthreads) parts <<= 1;
  std::vector<size_t> cut(parts + 1);
  for (int i = 0; i <= parts; i++) cut[i] = n * (size_t)i / parts;
  {
    std::vector<std::thread> th;
    for (int i = 0; i < parts; i++)
      th.emplace_back([&, i] { std::sort(v.begin() + cut[i], v.begin() + cut[i + 1], less); });
    for (auto& t : th) t.join();
  }
  for (int w = 1; w < parts; w <<= 1) {
    std::vector<std::thread> th;
    for (int i = 0; i + w < parts + 0; i += 2 * w)
      th.emplace_back([&, i, w] {
        std::inplace_merge(v.begin() + cut[i], v.begin() + cut[i + w],
                           v.begin() + cut[std::min(i + 2 * w, parts)], less);
      });
    for (auto& t : th) t.join();
  }
}

// cmd/muscato_confirm/main.go:161-164, 424-448
struct QRec {
  int mismatch;
  orc_hit hit;
};

static void qinsert(std::vector<QRec>& q, const QRec& a, int max_matches) {
  q.push_back(a);
  size_t ii = q.size() - 1;
  while (ii > 0) {
    size_t jj = (ii - 1) / 2;
    if (q[jj].mismatch > q[ii].mismatch) {
      std::swap(q[jj], q[ii]);
      ii = jj;
    } else {
      break;
    }
  }
  if ((int64_t)q.size() > (int64_t)max_matches) q.resize(max_matches);
}

// cmd/muscato_confirm/main.go:151-159
static inline int cdiff(const char* x, const char* y, int n) {
  int c = 0;
  for (int i = 0; i < n; i++) c += x[i] != y[i];
  return c;
}

// cmd/muscato_confirm/main.go:171-250 for one (window, key) block.
static void searchpairs(const Ctx& c, int k, const uint32_t* src, size_t nsrc, const Cand* mat,
                        size_t nmat, std::vector<orc_hit>& out) {
  const int q1 = c.P->windows[k], q2 = q1 + c.ww;
  const bool first = c.P->match_mode_first != 0;
  std::vector<QRec> qvals;
  for (size_t mi = 0; mi < nmat; mi++) {
    const Cand& m = mat[mi];
    const char* g = c.genes.ptr(m.gene);
    uint32_t jy, jz;
    cand_right(c, k, m, &jy, &jz);
    const int mlft_len = (m.pos == 0) ? 0 : q1;
    const char* mlft = g + m.pos - mlft_len;
    const int mrgt_len = (int)jz - (int)jy;
    for (size_t si = 0; si < nsrc; si++) {
      const uint32_t ri = src[si];
      const char* r = c.reads.ptr(ri);
      const int rl = (int)c.reads.len(ri);
      const int srgt_len = rl - q2;
      // :198  (len(stag)+len(slft)+len(srgt) == read length)
      const int nmiss = (int)((1 - c.P->pmatch) * (double)rl);
      if (srgt_len > mrgt_len) continue;  // :201-203
      // :207 cdiff(mlft, slft) iterates over len(mlft); slft has q1 bytes.
      int nx = cdiff(mlft, r, mlft_len);
      nx += cdiff(g + jy, r + q2, srgt_len);
      if (nx > nmiss) continue;
      QRec qq;
      qq.mismatch = nx;
      qq.hit = orc_hit{ri, m.gene, (uint32_t)(m.pos - mlft_len), (uint32_t)nx};  // :229
      if (first) {
        qvals.push_back(qq);
        if ((int64_t)qvals.size() > (int64_t)c.P->max_matches) goto done;  // :236-238
      } else {
        qinsert(qvals, qq, c.P->max_matches);
      }
    }
  }
done:
  for (auto& q : qvals) out.push_back(q.hit);
}

}  // namespace

extern "C" {

// timings[5]: window_reads+sort, bloom build, target scan, candidate sort, confirm (seconds)
// counts[3]:  total read windows, total candidates, total accepted pairs (before the union)
int orc_match(const char* reads, const uint64_t* roff, uint64_t nreads, const char* genes,
              const uint64_t* goff, uint64_t ngenes, const orc_params* P, orc_hit** out,
              uint64_t* nout, double* timings, uint64_t* counts) {
  const int W = P->n_windows, ww = P->window_width, NH = P->num_hash;
  if (W < 1 || W > ORC_MAX_WINDOWS || ww < 1 || NH < 1 || P->bloom_size == 0) return 1;
  Ctx c{Seqs{reads, roff, nreads}, Seqs{genes, goff, ngenes}, P, ww};
  const int nthr = P->nthreads > 0 ? P->nthreads : 1;
  double T[5] = {0, 0, 0, 0, 0};
  uint64_t C[3] = {0, 0, 0};

  // ---- muscato_window_reads (cmd/muscato_window_reads/main.go:94-141) + sort
  auto t0 = clk::now();
  std::vector<std::vector<uint32_t>> src(W);
  for (int k = 0; k < W; k++) {
    const int q1 = P->windows[k], q2 = q1 + ww;
    uint64_t nvalid = 0;
    for (uint64_t i = 0; i < nreads; i++) {
      if ((int)c.reads.len(i) < q2) continue;
      nvalid++;
      if (count_dinuc(c.reads.ptr(i) + q1, ww) < P->min_dinuc) continue;
      src[k].push_back((uint32_t)i);
    }
    // :143-151 -- "Window k produced no valid reads, exiting"
    if (nvalid == 0) return 2;
    parallel_sort(src[k], [&](uint32_t a, uint32_t b) { return cmp_src(c, k, a, b) < 0; }, nthr);
    C[0] += src[k].size();
  }
  auto t1 = clk::now();
  T[0] = secs(t0, t1);

  // ---- muscato_screen: buildBloom (cmd/muscato_screen/main.go:116-207)
  Tables tables(NH);
  std::vector<BitArray> smp;
  smp.reserve(W);
  for (int k = 0; k < W; k++) smp.emplace_back(P->bloom_size);
  {
    std::vector<std::thread> th;  // one worker per window (:136-162)
    for (int k = 0; k < W; k++)
      th.emplace_back([&, k] {
        const int q1 = P->windows[k];
        for (uint32_t ri : src[k]) {  // same rule as :174-185
          const unsigned char* s = (const unsigned char*)c.reads.ptr(ri) + q1;
          for (int j = 0; j < NH; j++) {
            const uint32_t* tb = tables.tab(j);
            uint32_t h = 0;
            for (int i = 0; i < ww; i++) h = rotl32(h, 1) ^ tb[s[i]];  // Reset+Write
            smp[k].set((uint64_t)h % P->bloom_size);                   // :155-156
          }
        }
      });
    for (auto& t : th) t.join();
  }
  auto t2 = clk::now();
  T[1] = secs(t1, t2);

  // ---- muscato_screen: search/processSeq (cmd/muscato_screen/main.go:256-366, 408-480)
  std::vector<std::vector<std::vector<Cand>>> cand_t(nthr, std::vector<std::vector<Cand>>(W));
  std::atomic<int> err{0};
  {
    std::atomic<uint64_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nthr; t++)
      th.emplace_back([&, t] {
        std::vector<uint32_t> hs(NH);
        std::vector<uint64_t> iw(NH);
        std::vector<int> ix;
        auto check_win = [&]() {  // :220-253
          for (int j = 0; j < NH; j++) iw[j] = (uint64_t)hs[j] % P->bloom_size;
          ix.clear();
          for (int k = 0; k < W; k++) {
            bool g = true;
            for (int j = 0; j < NH; j++)
              if (!smp[k].get(iw[j])) {
                g = false;
                break;
              }
            if (g) ix.push_back(k);
          }
        };
        for (;;) {
          uint64_t g0 = next.fetch_add(256);
          if (g0 >= ngenes) break;
          uint64_t g1 = std::min<uint64_t>(g0 + 256, ngenes);
          for (uint64_t gi = g0; gi < g1; gi++) {
            const unsigned char* seq = (const unsigned char*)c.genes.ptr(gi);
            const int len = (int)c.genes.len(gi);
            if (len < ww) continue;  // :268-271
            for (int j = 0; j < NH; j++) {
              const uint32_t* tb = tables.tab(j);
              uint32_t h = 0;
              for (int i = 0; i < ww; i++) h = rotl32(h, 1) ^ tb[seq[i]];
              hs[j] = h;
            }
            check_win();
            for (int i : ix) {  // :294-316
              const int q1 = P->windows[i];
              if (q1 != 0) continue;
              const int q2 = q1 + ww;
              int jz = 100 - q2;
              if (jz > len) jz = len;
              if (jz < ww) {  // Go: slice bounds out of range -> panic
                err = 3;
                return;
              }
              cand_t[t][i].push_back(Cand{(uint32_t)gi, 0});
            }
            for (int j = ww; j < len; j++) {  // :319-365
              for (int h = 0; h < NH; h++) {
                const uint32_t* tb = tables.tab(h);
                hs[h] = rotl32(hs[h], 1) ^ rotl32(tb[seq[j - ww]], ww) ^ tb[seq[j]];
              }
              check_win();
              for (int i : ix) {
                const int q1 = P->windows[i], q2 = q1 + ww;
                if (j < q2 - 1) continue;
                const int jx = j - ww + 1, jy = j + 1, jw = jx - q1;
                int jz = jy + P->max_read_length - q2;
                if (jz > len) jz = len;
                if (jz < jy) {  // Go: slice bounds out of range -> panic
                  err = 3;
                  return;
                }
                if (jw >= 0) cand_t[t][i].push_back(Cand{(uint32_t)gi, (uint32_t)jx});
              }
            }
          }
        }
      });
    for (auto& t : th) t.join();
  }
  if (err) return err;
  auto t3 = clk::now();
  T[2] = secs(t2, t3);

  // ---- sortBloom (cmd/muscato/main.go:318-385)
  std::vector<std::vector<Cand>> cand(W);
  for (int k = 0; k < W; k++) {
    size_t tot = 0;
    for (int t = 0; t < nthr; t++) tot += cand_t[t][k].size();
    cand[k].reserve(tot);
    for (int t = 0; t < nthr; t++) {
      cand[k].insert(cand[k].end(), cand_t[t][k].begin(), cand_t[t][k].end());
      std::vector<Cand>().swap(cand_t[t][k]);
    }
    parallel_sort(cand[k], [&](const Cand& a, const Cand& b) { return cmp_cand(c, k, a, b) < 0; },
                  nthr);
    C[1] += cand[k].size();
  }
  auto t4 = clk::now();
  T[3] = secs(t3, t4);

  // ---- muscato_confirm merge loop (cmd/muscato_confirm/main.go:357-416)
  std::vector<orc_hit> all;
  for (int k = 0; k < W; k++) {
    const int q1 = P->windows[k];
    const auto& S = src[k];
    const auto& M = cand[k];
    // Q12: the reference indexes recs[0] of an empty block and
    // panics when either file is empty; nothing can match then, emit nothing.
    if (S.empty() || M.empty()) continue;
    // block boundaries
    struct Blk { size_t s0, s1, m0, m1; };
    std::vector<Blk> blks;
    size_t si = 0, mi = 0;
    auto skey = [&](size_t i) { return c.reads.ptr(S[i]) + q1; };
    auto mkey = [&](size_t i) { return c.genes.ptr(M[i].gene) + M[i].pos; };
    while (si < S.size() && mi < M.size()) {
      int r = memcmp(skey(si), mkey(mi), ww);
      if (r == 0) {
        size_t s1 = si + 1, m1 = mi + 1;
        while (s1 < S.size() && memcmp(skey(s1), skey(si), ww) == 0) s1++;
        while (m1 < M.size() && memcmp(mkey(m1), mkey(mi), ww) == 0) m1++;
        blks.push_back(Blk{si, s1, mi, m1});
        si = s1;
        mi = m1;
      } else if (r < 0) {
        si++;
      } else {
        mi++;
      }
    }
    std::vector<std::vector<orc_hit>> outs(nthr);
    std::atomic<size_t> nb{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nthr; t++)
      th.emplace_back([&, t] {
        for (;;) {
          size_t b0 = nb.fetch_add(64);
          if (b0 >= blks.size()) break;
          size_t b1 = std::min(b0 + 64, blks.size());
          for (size_t b = b0; b < b1; b++)
            searchpairs(c, k, &S[blks[b].s0], blks[b].s1 - blks[b].s0, &M[blks[b].m0],
                        blks[b].m1 - blks[b].m0, outs[t]);
        }
      });
    for (auto& t : th) t.join();
    for (auto& o : outs) all.insert(all.end(), o.begin(), o.end());
  }
  C[2] = all.size();
  auto t5 = clk::now();
  T[4] = secs(t4, t5);

  // ---- combine_filter + sort -u at the tuple level (cmd/muscato/main.go:441-463).
  // With unique read sequences, distinct tuples <=> distinct rmatch lines.
  auto lt = [](const orc_hit& a, const orc_hit& b) {
    if (a.read_idx != b.read_idx) return a.read_idx < b.read_idx;
    if (a.gene_idx != b.gene_idx) return a.gene_idx < b.gene_idx;
    if (a.pos != b.pos) return a.pos < b.pos;
    return a.nmiss < b.nmiss;
  };
  std::sort(all.begin(), all.end(), lt);
  all.erase(std::unique(all.begin(), all.end(),
                        [](const orc_hit& a, const orc_hit& b) {
                          return a.read_idx == b.read_idx && a.gene_idx == b.gene_idx &&
                                 a.pos == b.pos && a.nmiss == b.nmiss;
                        }),
            all.end());

  *nout = all.size();
  *out = (orc_hit*)malloc(sizeof(orc_hit) * (all.size() ? all.size() : 1));
  if (!*out) return 4;
  if (!all.empty()) memcpy(*out, all.data(), sizeof(orc_hit) * all.size());
  if (timings) memcpy(timings, T, sizeof T);
  if (counts) memcpy(counts, C, sizeof C);
  return 0;
}

void orc_free(orc_hit* p) { free(p); }

int orc_count_dinuc(const char* s, int n) { return count_dinuc(s, n); }

}  // extern "C"
