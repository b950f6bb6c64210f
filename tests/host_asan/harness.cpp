// CPU sanitizer leg of libg3hip's host logic ("-fsanitize=address host build of the C-ABI
// shim"): g3py_amd/csrc/g3_host.h is pure C++ and is compiled here with g++ -fsanitize=address,undefined.  Every
// table / schedule builder is driven over the shapes the library produces (and some it never should) and its output
// is checked the way the device consumes it: the tile lookup of the GEMM kernel, the op list of the stripe solve, the
// chunking of the multi-GPU staircase, panel boundaries, the fast-path matcher, the program ring, the jitter schedule.
// TEST INFRASTRUCTURE: never linked into the product.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <random>
#include <set>
#include <vector>

#include "g3_host.h"

static int g_fail = 0;
#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) {                                                         \
      fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
      if (++g_fail > 20) exit(1);                                       \
    }                                                                   \
  } while (0)

// the device side of the raster: virtual tile id -> (row tile, column tile), as gemm_nt_kernel does it
static void lookup(const RasterTab& tab, int v, int* bm, int* bn) {
  int lo = 0, hi = tab.ngroups;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tab.g[mid].prefix <= v) lo = mid; else hi = mid;
  }
  const int w = v - tab.g[lo].prefix, rows = (int)tab.g[lo].nrows;
  *bn = w / rows;
  *bm = (int)tab.g[lo].row0 + (w - *bn * rows);
}

template <int BM, int BN>
static void check_raster(const GemmShape& sh) {
  RasterTab tab;
  memset(&tab, 0xCD, sizeof(tab));
  const long long nv = build_raster<BM, BN>(sh, &tab);
  if (nv < 0) {   // refused: only legal for very long staircases / block tables
    CHECK(sh.kind == 2 || sh.m / BM > 65535LL * G3_RASTER_MAX);
    return;
  }
  CHECK(tab.ngroups >= 0 && tab.ngroups <= G3_RASTER_MAX);
  CHECK(tab.g[tab.ngroups].prefix == (int)nv);
  for (int g = 0; g + 1 <= tab.ngroups; ++g) CHECK(tab.g[g].prefix <= tab.g[g + 1].prefix);
  // every launched tile: inside C, unique, and (for the wanted region) complete
  std::set<std::pair<int, int>> seen;
  const int64_t tiles_m = sh.m / BM;
  for (long long v = 0; v < nv; ++v) {
    int bm, bn;
    lookup(tab, (int)v, &bm, &bn);
    CHECK(bm >= 0 && bm < tiles_m);
    CHECK(bn >= 0 && (int64_t)bn * BN < sh.n);
    CHECK(seen.insert({bm, bn}).second);
  }
  // coverage: every wanted element's tile was launched
  if (sh.kind == 0) {
    CHECK((long long)seen.size() == tiles_m * (sh.n / BN));
  } else if (sh.kind == 1) {
    for (int64_t i = 0; i < sh.m; i += 61)
      for (int64_t j = 0; j < sh.n; j += 53)
        if (j <= i + sh.off) CHECK(seen.count({(int)(i / BM), (int)(j / BN)}) == 1);
    // and no tile that lies entirely above the shifted diagonal of its group's last row
    for (auto& t : seen) {
      const int64_t last_row_of_group_at_most = ((int64_t)t.first / GROUP_M + 1) * GROUP_M * BM + (int64_t)GROUP_M * BM;
      CHECK((int64_t)t.second * BN <= last_row_of_group_at_most + sh.off + (int64_t)65536 * BM);
    }
  } else {
    int64_t r = 0;
    for (int s = 0; s < sh.nseg; ++s) {
      const bool dg = sh.seg_diag && sh.seg_diag[s];
      const int64_t c0 = sh.seg_cols[s] - sh.seg_rows[s];
      for (int64_t i = 0; i < sh.seg_rows[s]; i += 97)
        for (int64_t j = 0; j < sh.seg_cols[s]; j += 89) {
          const bool wanted = !dg || j <= c0 + i;
          if (wanted) CHECK(seen.count({(int)((r + i) / BM), (int)(j / BN)}) == 1);
        }
      // nothing beyond the segment's width
      for (auto& t : seen)
        if ((int64_t)t.first * BM >= r && (int64_t)t.first * BM < r + sh.seg_rows[s]) CHECK((int64_t)t.second * BN < sh.seg_cols[s]);
      r += sh.seg_rows[s];
    }
    if (sh.b_nb > 0 && sh.b_perm) {
      CHECK(tab.b_nb == (int)sh.b_nb);
      for (int i = 0; i < sh.nperm; ++i) CHECK(tab.b_blk[i] == (unsigned short)sh.b_perm[i]);
    } else {
      CHECK(tab.b_nb == 0);
    }
  }
  // the algorithmic element count is what the profiler divides by: compare with a brute-force count
  double cnt = 0;
  if (sh.kind == 0) cnt = (double)sh.m * sh.n;
  else if (sh.kind == 1) { for (int64_t i = 0; i < sh.m; ++i) { int64_t c = i + sh.off + 1; c = c < 0 ? 0 : (c > sh.n ? sh.n : c); cnt += c; } }
  else for (int s = 0; s < sh.nseg; ++s) {
    cnt += (double)sh.seg_rows[s] * sh.seg_cols[s];
    if (sh.seg_diag && sh.seg_diag[s] && sh.seg_cols[s] >= sh.seg_rows[s]) cnt -= 0.5 * (double)sh.seg_rows[s] * (sh.seg_rows[s] - 1.0);
  }
  CHECK(fabs(shape_elems(sh) - cnt) <= 1e-9 * (cnt + 1));
}

static void test_rasters() {
  std::mt19937 rng(5);
  { GemmShape big{1, 65536 + 128, 1024, 0, 0, nullptr, nullptr, 0, nullptr, 0, nullptr}; check_raster<128, 128>(big); }
  for (int64_t m : {64, 128, 1024, 4096, 30720})
    for (int64_t n : {64, 128, 1024, 4096}) {
      GemmShape d{0, m, n, 0, 0, nullptr, nullptr, 0, nullptr, 0, nullptr};
      check_raster<64, 64>(d);
      if (m % 128 == 0 && n % 128 == 0) check_raster<128, 128>(d);
      for (int64_t off : {(int64_t)0, (int64_t)-128, (int64_t)256, m, -m, (int64_t)1 << 20}) {
        GemmShape t{1, m, n, off, 0, nullptr, nullptr, 0, nullptr, 0, nullptr};
        check_raster<64, 64>(t);
        if (m % 128 == 0 && n % 128 == 0) check_raster<128, 128>(t);
        if (m % 32 == 0 && n % 128 == 0) check_raster<32, 128>(t);
      }
    }
  // staircases: random segments, block tables, diagonal flags, more segments than one table can hold
  for (int it = 0; it < 80; ++it) {
    const int nseg = 1 + (int)(rng() % (it % 10 == 0 ? 260 : 24));
    const int64_t nb = 128 * (1 + rng() % 8);
    std::vector<int64_t> rows(nseg), cols(nseg), diag(nseg);
    int64_t width = 0;
    for (int s = 0; s < nseg; ++s) {
      rows[s] = (rng() % 9 == 0) ? 0 : nb;
      cols[s] = nb * (rng() % (it % 10 == 0 ? 200 : 24));
      diag[s] = (rng() % 2) && cols[s] >= rows[s];
      width = cols[s] > width ? cols[s] : width;
    }
    if (width == 0) continue;
    const int nperm = (int)(width / nb);
    std::vector<int32_t> perm(nperm);
    for (int i = 0; i < nperm; ++i) perm[i] = (int32_t)(rng() % 4096);
    int64_t m = 0;
    for (auto r : rows) m += r;
    if (m == 0) continue;
    const bool with_perm = rng() % 2;
    GemmShape sh{2, m, width, 0, nseg, rows.data(), cols.data(), with_perm ? nb : 0, with_perm ? perm.data() : nullptr,
                 with_perm ? nperm : 0, (rng() % 2) ? diag.data() : nullptr};
    check_raster<128, 128>(sh);
    check_raster<64, 64>(sh);
    // and through the chunker of the multi-GPU driver: the chunks tile the staircase exactly once
    std::vector<G3hStairChunk> ch;
    const int limit = g3h_tune_from_env().stair_max;        // (the test lowers G3_STAIR_MAX)
    g3h_stair_chunks(rows, cols, nb, nperm, &ch, sh.seg_diag ? &diag : nullptr, limit);
    std::map<std::pair<int64_t, int64_t>, int> cover;   // (segment-row block, column block) -> times covered
    for (auto& c : ch) {
      CHECK((int)c.rows.size() <= limit && c.nblk <= limit && c.nblk >= 1);
      CHECK(c.blk0 >= 0 && c.blk0 + c.nblk <= nperm);
      CHECK(c.col0 % nb == 0 && c.col0 == (int64_t)c.blk0 * nb);
      int64_t r = c.row0;
      for (size_t s = 0; s < c.rows.size(); ++s) {
        CHECK(c.cols[s] >= 0 && c.cols[s] <= (int64_t)c.nblk * nb);
        if (c.diag[s]) CHECK(c.cols[s] >= c.rows[s]);
        if (c.rows[s] > 0)
          for (int64_t j = 0; j < c.cols[s]; j += nb) cover[{r, c.col0 + j}]++;
        r += c.rows[s];
      }
      // what the launch itself will build
      GemmShape sub{2, 0, 0, 0, (int)c.rows.size(), c.rows.data(), c.cols.data(), nb, perm.data() + c.blk0, c.nblk,
                    sh.seg_diag ? c.diag.data() : nullptr};
      for (size_t s = 0; s < c.rows.size(); ++s) { sub.m += c.rows[s]; sub.n = c.cols[s] > sub.n ? c.cols[s] : sub.n; }
      if (sub.m > 0 && sub.n > 0) check_raster<128, 128>(sub);
    }
    int64_t r = 0;
    for (int s = 0; s < nseg; ++s) {
      if (rows[s] > 0)
        for (int64_t j = 0; j < cols[s]; j += nb) CHECK((cover[{r, j}] == 1));
      r += rows[s];
    }
    size_t want = 0;
    for (int s = 0; s < nseg; ++s) if (rows[s] > 0) want += (size_t)(cols[s] / nb);
    CHECK(cover.size() == want);
    // exactly one chunk carries each flagged diagonal block
    if (sh.seg_diag) {
      std::vector<int> got(nseg, 0);
      for (auto& c : ch) {
        // segments of a chunk are consecutive segments of the staircase starting at the one whose first row is c.row0
        int s0 = 0; int64_t rr = 0;
        while (s0 < nseg && rr < c.row0) rr += rows[s0++];
        while (s0 < nseg && rows[s0] == 0 && rr == c.row0 && s0 % limit != 0) ++s0;
        for (size_t s = 0; s < c.rows.size() && s0 + (int)s < nseg; ++s) got[(s0 / limit) * limit + s] += c.diag[s] ? 1 : 0;
      }
      for (int s = 0; s < nseg; ++s) if (rows[s] > 0 && diag[s]) CHECK(got[s] == 1);
    }
  }
}

static void test_trsm_ops() {
  // run the op list on the host (one "stripe" of 3 rows) and compare with a direct triangular solve
  std::mt19937 rng(7);
  std::uniform_real_distribution<double> U(-1, 1);
  for (int64_t n = 128; n <= 1024; n += 128) {
    TrsmOps ops;
    memset(&ops, 0xCD, sizeof(ops));
    ops.nops = 0;
    trsm_ops_rec(&ops, 0, n);
    CHECK(ops.nops >= 1 && ops.nops <= G3_TRSM_MAXOPS);
    std::vector<double> L(n * n, 0.0), X(3 * n), X0;
    for (int64_t i = 0; i < n; ++i) { for (int64_t j = 0; j < i; ++j) L[i * n + j] = 0.05 * U(rng); L[i * n + i] = 1.0 + 0.5 * fabs(U(rng)); }
    for (auto& v : X) v = U(rng);
    X0 = X;
    // inverses of the 128 x 128 diagonal blocks
    std::vector<double> W((n / 128) * 128 * 128, 0.0);
    for (int64_t b = 0; b < n / 128; ++b)
      for (int c = 0; c < 128; ++c) {          // column c of inv(L_bb) by forward substitution
        double col[128];
        for (int i = 0; i < 128; ++i) {
          double s = (i == c) ? 1.0 : 0.0;
          for (int j = 0; j < i; ++j) s -= L[(b * 128 + i) * n + b * 128 + j] * col[j];
          col[i] = s / L[(b * 128 + i) * n + b * 128 + i];
        }
        for (int i = 0; i < 128; ++i) W[b * 128 * 128 + i * 128 + c] = col[i];
      }
    for (int q = 0; q < ops.nops; ++q) {
      const auto o = ops.op[q];
      CHECK(o.col >= 0 && o.col + 128 <= n && o.acol >= 0 && o.acol + o.k <= n && o.k > 0 && o.k % 128 == 0);
      for (int r = 0; r < 3; ++r) {
        double out[128];
        for (int c = 0; c < 128; ++c) {
          double s = 0;
          if (o.leaf) {     // X_j <- X_j W_j^T
            CHECK(o.brow >= 0 && o.brow < n / 128 && o.k == 128 && o.acol == o.col);
            for (int t = 0; t < 128; ++t) s += X[r * n + o.acol + t] * W[o.brow * 128 * 128 + c * 128 + t];
            out[c] = s;
          } else {          // X_hi -= X_lo L_hi,lo^T
            CHECK(o.brow == o.col && o.bcol == o.acol && o.acol + o.k <= o.col);
            for (int t = 0; t < o.k; ++t) s += X[r * n + o.acol + t] * L[(o.brow + c) * n + o.bcol + t];
            out[c] = X[r * n + o.col + c] - s;
          }
        }
        for (int c = 0; c < 128; ++c) X[r * n + o.col + c] = out[c];
      }
    }
    // X L^T must equal X0
    double err = 0;
    for (int r = 0; r < 3; ++r)
      for (int64_t j = 0; j < n; ++j) {
        double s = 0;
        for (int64_t t = 0; t <= j; ++t) s += X[r * n + t] * L[j * n + t];
        err = fmax(err, fabs(s - X0[r * n + j]));
      }
    CHECK(err < 1e-10);
  }
}

static void test_panel_bounds() {
  for (int64_t n = 128; n <= 70000; n += (n < 2048 ? 128 : 11 * 128 + (n % 1024)))
    for (int64_t NB : {128, 256, 512, 1024, 2048})
      for (int G : {1, 2, 3, 8})
        for (int batch : {1, 16}) {
          const int64_t np = g3h_roundup(n, 128);
          std::vector<int64_t> b;
          std::vector<int> g;
          g3h_panel_bounds(np, NB, G, batch, g3h_tune_from_env(), &b, &g);
          CHECK(b.size() >= 2 && b.front() == 0 && b.back() == np);
          for (size_t i = 0; i + 1 < b.size(); ++i) {
            CHECK(b[i] < b[i + 1] || (i + 2 == b.size() && b[i] <= b[i + 1]));
            CHECK(b[i] % 128 == 0);
            CHECK(b[i + 1] - b[i] <= g3h_roundup(NB, 128) || i + 2 == b.size());
          }
          CHECK(g.front() == 0 && g.back() == (int)b.size() - 1);
          for (size_t i = 0; i + 1 < g.size(); ++i) CHECK(g[i] < g[i + 1] && g[i + 1] - g[i] <= G);
        }
}

static void test_match_and_validate() {
  std::mt19937 rng(11);
  for (int it = 0; it < 6000; ++it) {
    g3_kernel_prog p;
    memset(&p, 0, sizeof(p));
    const bool wild = it % 5 == 0;       // out-of-range structure must be refused by validate before any matching
    p.nleaf = wild ? (int)(rng() % 12) - 2 : 1 + (int)(rng() % 3);
    p.nprod = wild ? (int)(rng() % 20) - 2 : 1 + (int)(rng() % 3);
    const int d = 1 + (int)(rng() % 8);
    for (int l = 0; l < G3_MAXLEAF; ++l) {
      p.leaf[l].kind = wild ? (int)(rng() % 14) - 2 : (int)(rng() % 11);
      p.leaf[l].ndims = wild ? (int)(rng() % 40) - 2 : d;
      for (int k = 0; k < G3_MAXD; ++k) { p.leaf[l].dims[k] = wild ? (int)(rng() % 50) - 5 : (k < d ? k : 0); p.leaf[l].rate[k] = 1.0; p.leaf[l].freq[k] = 0.3; }
      p.leaf[l].var = 1.0; p.leaf[l].alpha = 2.0;
    }
    for (int q = 0; q < G3_MAXPROD; ++q) {
      p.prod[q].coef = (rng() % 7 == 0) ? 2.0 : 1.0;
      p.prod[q].nfac = wild ? (int)(rng() % 7) - 1 : 1 + (int)(rng() % 2 == 0 ? 0 : 1);
      for (int f = 0; f < G3_MAXFAC; ++f) p.prod[q].fac[f] = wild ? (int)(rng() % 12) - 2 : (int)(rng() % (p.nleaf > 0 ? p.nleaf : 1));
    }
    if (g3h_validate_prog(&p, d)) continue;          // the library stops here (status -2)
    int pk;
    SeParams<double, 1> s1; SeParams<double, 2> s2; SeParams<double, 3> s3; SeParams<double, 4> s4; SeParams<double, 8> s8;
    SeParams<float, 16> s16;
    int k = -1;
    k = g3h_match_fast<double, 1>(&p, d, &s1, &pk);
    if (k < 0) k = g3h_match_fast<double, 2>(&p, d, &s2, &pk);
    if (k < 0) k = g3h_match_fast<double, 3>(&p, d, &s3, &pk);
    if (k < 0) k = g3h_match_fast<double, 4>(&p, d, &s4, &pk);
    if (k < 0) k = g3h_match_fast<double, 8>(&p, d, &s8, &pk);
    if (k < 0) k = g3h_match_fast<float, 16>(&p, d, &s16, &pk);
    if (k >= 0) {
      CHECK(k == G3_K_SE || k == G3_K_OU || k == G3_K_MAT32 || k == G3_K_MAT52 || k == G3_K_RQ);
      CHECK(pk == -1 || ((pk == G3_K_COS || pk == G3_K_SIN || pk == G3_K_SM) && (k == G3_K_SE || k == G3_K_MAT32 || k == G3_K_MAT52) && (d == 1 || d == 2 || d == 4 || d == 8)));
      CHECK(p.nprod <= 3 && p.shift == 0.0);
      int two = 0;
      for (int q = 0; q < p.nprod; ++q) { CHECK((p.prod[q].nfac == 1 || p.prod[q].nfac == 2) && p.prod[q].coef == 1.0); two += p.prod[q].nfac == 2; }
      CHECK(two <= 1 && (two == 0 || pk >= 0));      // a two-factor term is stationary * periodic, at most one
    }
  }
}

static void test_ring_and_jitter() {
  bool busy[8] = {false};
  int next = 0, last = -1;
  std::vector<int> order;
  for (int it = 0; it < 100; ++it) {
    bool wait; int mark;
    const int s = g3h_ring_take(busy, 8, &next, &last, &wait, &mark);
    CHECK(s >= 0 && s < 8 && s == it % 8);
    CHECK(mark == (it == 0 ? -1 : (it - 1) % 8));
    CHECK(wait == (it >= 8));          // a slot is only waited for once the ring has wrapped onto it
    CHECK(last == s && !busy[s]);
  }
  // the reference's schedule: dK = mean * 1e-6 (float32 constant), x10 per retry, lift when min <= 0
  G3hJitter j(2.0, 0.5);
  CHECK(j.lift == 0.0 && fabs(j.value() - 2.0 * (double)1e-6f) < 1e-18 && j.usable());
  double prev = j.value();
  for (int t = 0; t < 19; ++t) { j.next(); CHECK(fabs(j.value() / prev - 10.0) < 1e-12); prev = j.value(); }
  CHECK(j.tries == 19 && G3hJitter::max_tries() == 20);
  G3hJitter k(3.0, -0.25);
  CHECK(fabs(k.lift - (3.0 * (double)1e-6f + 0.25)) < 1e-15 && k.value() > 0.25);
  G3hJitter n(NAN, 1.0);
  CHECK(!n.usable());
}

static void test_dealing() {
  // the multi-GPU driver's block dealing and gather tables: every block has exactly one owner, the rounds keep the
  // counts within one and the ranks' total work (block I weighs I^2 + 6 I + 1) within a few per cent once every rank holds
  // four blocks, the table is injective, rank-major, ascending inside a rank, and never reaches past P * count
  for (int P = 1; P <= 9; ++P)
    for (int nblk = 1; nblk <= 70; nblk += (nblk < 20 ? 1 : 7)) {
      std::vector<int> cnt(P, 0), owner;
      std::vector<double> load(P, 0.0);
      g3h_deal(P, nblk, &owner);
      CHECK((int)owner.size() == nblk);
      for (int I = 0; I < nblk; ++I) {
        const int q = owner[I];
        CHECK(q >= 0 && q < P);
        cnt[q]++;
        load[q] += (double)I * I + 6.0 * I + 1.0;
      }
      int lo = cnt[0], hi = cnt[0];
      double lmax = 0, lsum = 0;
      for (int q = 0; q < P; ++q) { lo = cnt[q] < lo ? cnt[q] : lo; hi = cnt[q] > hi ? cnt[q] : hi; lmax = load[q] > lmax ? load[q] : lmax; lsum += load[q]; }
      CHECK(hi - lo <= 1);
      if (nblk >= 4 * P) CHECK(lmax <= 1.05 * lsum / P);
      for (int a = -1; a < nblk; a += (nblk < 12 ? 1 : 5))
        for (int b = a; b < nblk; b += (nblk < 12 ? 1 : 3)) {
          const int first = a + 1, last = b;          // perm_of(k): k + 1 .. nblk - 1; perm_upto(k): 0 .. k
          std::vector<int32_t> idx;
          const int c = g3h_gather_table(owner, P, first, last, &idx);
          CHECK((int)idx.size() == (last >= first ? last - first + 1 : 0));
          std::vector<int> seen(P * (c > 0 ? c : 1), 0), prev(P, -1);
          for (int I = first; I <= last; ++I) {
            const int t = idx[I - first], q = owner[I];
            CHECK(c > 0 && t >= q * c && t < (q + 1) * c);      // inside the owner's slots
            CHECK(!seen[t]);                                    // injective
            seen[t] = 1;
            CHECK(t > prev[q]);                                 // ascending inside a rank
            prev[q] = t;
          }
        }
    }
}

int main(int argc, char** argv) {
  if (argc == 4 && !strcmp(argv[1], "deal")) {        // "deal P nblk": print the table (the Python twin must deal identically)
    std::vector<int> owner;
    g3h_deal(atoi(argv[2]), atoi(argv[3]), &owner);
    for (int q : owner) printf("%d ", q);
    printf("\n");
    return 0;
  }
  test_dealing();
  test_rasters();
  test_trsm_ops();
  test_panel_bounds();
  test_match_and_validate();
  test_ring_and_jitter();
  if (g_fail) { fprintf(stderr, "%d checks failed\n", g_fail); return 1; }
  printf("host_asan ok\n");
  return 0;
}
