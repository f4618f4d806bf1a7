// CPU sanitizer harness for the threaded HOST code of libbulklmm_hip.so (tests/sanitize/Makefile: `make check`):
//   host_path.hip   the pinned staging ring and the CopyPool that moves L into pageable caller memory
//   blmm_multi.hip  one worker thread per device, job hand-off, every gather mode, the caller's current device restored
//   readers.hip     the CSV / Helium parsers (malformed and ragged input)
// compiled as plain C++ against tests/sanitize/hip/hip_runtime.h (a CPU stand-in for the HIP runtime) and a fake compute core
// below (blmm_create / blmm_bulkscan / ... with a trivial "scan" on the CPU), under -fsanitize=address,undefined and, as a
// second binary, -fsanitize=thread.
#include "../../bulklmm.jl_amd/csrc/blmm_internal.h"
#include <cstdio>
#include <cmath>
#include <string>
#include <vector>

thread_local int stub_current_device = 0;
int stub_device_count = 2;

// ---- the fake core: what blmm_multi.hip and host_path.hip link against instead of blmm_api.hip + the kernels -------------------
namespace blmm {
int fail(blmm_ctx* ctx, int code, const std::string& msg) { if (ctx) ctx->err = msg; return code; }
int ensure(blmm_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes == 0) bytes = 8;
  if (b.cap >= bytes) return BLMM_OK;
  std::free(b.p);
  b.p = std::malloc(bytes); b.cap = bytes;
  return b.p ? BLMM_OK : fail(ctx, BLMM_ERR_ALLOC, "malloc");
}
// CPU stand-ins for the two column-pass launchers the multi-device consumers call (kernels_prep.hip / kernels_post.hip)
int launch_colmax(blmm_ctx*, const double* L, int64_t p, int64_t m, int64_t ldL, double* mx, int64_t* arg) {
  for (int64_t j = 0; j < m; ++j) {
    double best = -INFINITY; int64_t bi = -1;
    for (int64_t i = 0; i < p; ++i) if (L[j * ldL + i] > best) { best = L[j * ldL + i]; bi = i; }
    mx[j] = best; if (arg) arg[j] = bi;
  }
  return BLMM_OK;
}
int launch_threshold(blmm_ctx*, const double* L, int64_t p, int64_t m, int64_t ldL, double thr, int64_t cap, int32_t* di, int32_t* dj,
                     double* dl, int64_t* dcount) {
  int64_t c = 0;
  for (int64_t j = 0; j < m; ++j)
    for (int64_t i = 0; i < p; ++i)
      if (L[j * ldL + i] > thr) { if (c < cap) { di[c] = (int32_t)i; dj[c] = (int32_t)j; dl[c] = L[j * ldL + i]; } ++c; }
  *dcount = c;
  return BLMM_OK;
}
}  // namespace blmm
using namespace blmm;

static void fake_scan(const double* Y, int64_t n, int64_t m, const double* G, int64_t p, double* L, int64_t ldL, double* h2, bool alt) {
  for (int64_t j = 0; j < m; ++j) {
    for (int64_t i = 0; i < p; ++i) {
      double s = 0.0;
      for (int64_t k = 0; k < n; ++k) s += Y[j * n + k] * G[i * n + k];
      L[j * ldL + i] = s;
      if (alt) h2[j * p + i] = 0.5 * s;
    }
    if (!alt) h2[j] = Y[j * n];
  }
}

extern "C" {
int blmm_device_count(void) { return stub_device_count; }
const char* blmm_err_string(int) { return "stub"; }
const char* blmm_last_error(const blmm_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }
int blmm_create(int device_id, void*, blmm_ctx** out) {
  if (device_id < 0 || device_id >= stub_device_count) return BLMM_ERR_NO_DEVICE;
  (void)hipSetDevice(device_id);
  blmm_ctx* c = new blmm_ctx(); c->device = device_id; *out = c; return BLMM_OK;
}
void blmm_destroy(blmm_ctx* ctx) {
  if (!ctx) return;
  std::free(ctx->outL.p); std::free(ctx->outH2.p); std::free(ctx->tmpA.p); std::free(ctx->tmpB.p); std::free(ctx->redtrip.p);
  destroy_host_stage(ctx->hstage);
  delete ctx;
}
int blmm_synchronize(blmm_ctx*) { return BLMM_OK; }
int blmm_bulkscan_dev(blmm_ctx* ctx, const blmm_opts* o, const double* dY, int64_t n, int64_t m, const double* dG, int64_t p, const double*,
                      int64_t, const double*, const double*, const double*, int64_t, double* dL, int64_t ldL, double* dh2, blmm_status* st) {
  (void)hipSetDevice(ctx->device);
  fake_scan(dY, n, m, dG, p, dL, ldL, dh2, o->method == BLMM_ALT_GRID);
  if (st) { std::memset(st, 0, sizeof(*st)); st->lowrank_rank = ctx->device + 1; }
  return BLMM_OK;
}
int blmm_bulkscan(blmm_ctx* ctx, const blmm_opts* o, const double* Y, int64_t n, int64_t m, const double* G, int64_t p, const double* C,
                  int64_t nc, const double* K, const double* w, const double* grid, int64_t ng, double* L_out, double* h2_out, blmm_status* st) {
  const bool alt = o->method == BLMM_ALT_GRID;
  int rc;
  if ((rc = ensure(ctx, ctx->outL, sizeof(double) * (size_t)(p * m + 1)))) return rc;
  if ((rc = ensure(ctx, ctx->outH2, sizeof(double) * (size_t)((alt ? p * m : m) + 1)))) return rc;
  rc = blmm_bulkscan_dev(ctx, o, Y, n, m, G, p, C, nc, K, w, grid, ng, ptr<double>(ctx->outL), p, ptr<double>(ctx->outH2), st);
  if (rc) return rc;
  ctx->last_L = ptr<double>(ctx->outL); ctx->last_p = p; ctx->last_m = m; ctx->last_f32 = false;
  // through the real device -> host leg (staging ring + CopyPool when the block is large); L_out == NULL: the block stays "in HBM"
  if (L_out && p * m > 0 && (rc = copy_to_host(ctx, L_out, ctx->outL.p, sizeof(double) * (size_t)(p * m)))) return rc;
  if (h2_out && (rc = copy_to_host(ctx, h2_out, ctx->outH2.p, sizeof(double) * (size_t)(alt ? p * m : m)))) return rc;
  return BLMM_OK;
}
}  // extern "C"

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

static void test_copy_pool() {
  blmm_ctx* ctx = nullptr;
  REQUIRE(blmm_create(0, nullptr, &ctx) == BLMM_OK);
  const size_t sizes[] = {0, 1, 4095, 4097, ((size_t)8 << 20) - 1, ((size_t)8 << 20) + 3, ((size_t)40 << 20) + 12345, ((size_t)140 << 20) + 7};
  setenv("BLMM_DEV_ENV", "1", 1);                                 // the library reads BLMM_* switches only as developer switches
  for (const char* nt : {"1", "2", "5", "9"}) {
    setenv("BLMM_D2H_THREADS", nt, 1);
    destroy_host_stage(ctx->hstage); ctx->hstage = nullptr;       // a fresh pool with this thread count
    for (size_t b : sizes) {
      if (b > ((size_t)100 << 20) && nt[0] != '5') continue;      // the ring wraps (more than 4 pieces of 32 MB): once is enough
      std::vector<unsigned char> src(b + 1), dst(b + 64, 0xAB);
      for (size_t i = 0; i < b; ++i) src[i] = (unsigned char)((i * 2654435761u) >> 13);
      REQUIRE(copy_to_host(ctx, dst.data() + 8, src.data(), b) == BLMM_OK);
      REQUIRE(std::memcmp(dst.data() + 8, src.data(), b) == 0);
      for (int g = 0; g < 8; ++g) { REQUIRE(dst[g] == 0xAB); REQUIRE(dst[8 + b + g] == 0xAB); }   // nothing outside the block
    }
  }
  blmm_destroy(ctx);
}

static void test_multi() {
  const int64_t n = 7, p = 33;
  blmm_opts o; std::memset(&o, 0, sizeof(o));
  const int devs4[4] = {0, 0, 1, 1}, devs3[3] = {1, 0, 1};
  for (int cfg = 0; cfg < 2; ++cfg) {
    blmm_multi* mc = nullptr;
    (void)hipSetDevice(cfg);                                        // the caller's current device must survive every call
    REQUIRE(blmm_create_multi(cfg ? devs3 : devs4, cfg ? 3 : 4, &mc) == BLMM_OK);
    int cur = -1; (void)hipGetDevice(&cur); REQUIRE(cur == cfg);
    const int R = blmm_multi_ndev(mc);
    for (int64_t m : {0, 1, 5, 64, 1001}) {
      std::vector<double> Y((size_t)(n * m + 1)), G((size_t)(n * p)), K((size_t)(n * n), 0.0);
      for (size_t i = 0; i < Y.size(); ++i) Y[i] = std::sin(0.37 * (double)i);
      for (size_t i = 0; i < G.size(); ++i) G[i] = std::cos(0.11 * (double)i);
      for (int method : {BLMM_NULL_GRID, BLMM_ALT_GRID}) {
        o.method = method;
        const bool alt = method == BLMM_ALT_GRID;
        std::vector<double> Lref((size_t)(p * m + 1)), Href((size_t)((alt ? p * m : m) + 1));
        fake_scan(Y.data(), n, m, G.data(), p, Lref.data(), p, Href.data(), alt);
        for (int gather : {BLMM_GATHER_HOST_SHARDS, BLMM_GATHER_NONE, BLMM_GATHER_ALLGATHER}) {
          for (int rep = 0; rep < 6; ++rep) {
            blmm_multi_opts mo; mo.gather_mode = gather; mo.reserved = 0;
            std::vector<double> L((size_t)(p * m + 1), -1.0), H((size_t)((alt ? p * m : m) + 1), -1.0);
            std::vector<blmm_status> st((size_t)R);
            REQUIRE(blmm_bulkscan_multi(mc, &o, &mo, Y.data(), n, m, G.data(), p, nullptr, 0, K.data(), nullptr, nullptr, 0, L.data(), H.data(), st.data()) == BLMM_OK);
            REQUIRE(std::memcmp(L.data(), Lref.data(), sizeof(double) * (size_t)(p * m)) == 0);
            REQUIRE(std::memcmp(H.data(), Href.data(), sizeof(double) * (size_t)(alt ? p * m : m)) == 0);
            (void)hipGetDevice(&cur); REQUIRE(cur == cfg);
            if (rep == 0 && m > 0) {
              // the consumers of the blocks where they are, in every gather mode -- and with host_shards also without any L_out
              std::vector<double> mx((size_t)m), mref((size_t)m); std::vector<int64_t> ax((size_t)m), aref((size_t)m);
              launch_colmax(nullptr, Lref.data(), p, m, p, mref.data(), aref.data());
              if (gather == BLMM_GATHER_HOST_SHARDS)
                REQUIRE(blmm_bulkscan_multi(mc, &o, &mo, Y.data(), n, m, G.data(), p, nullptr, 0, K.data(), nullptr, nullptr, 0, nullptr, H.data(), st.data()) == BLMM_OK);
              REQUIRE(blmm_multi_last_colmax(mc, mx.data(), ax.data()) == BLMM_OK);
              REQUIRE(mx == mref && ax == aref);
              const int64_t cap = 17;
              std::vector<int32_t> ti((size_t)cap), tj((size_t)cap); std::vector<double> tl((size_t)cap); int64_t cnt = -1, cref = 0;
              for (int64_t e = 0; e < p * m; ++e) cref += Lref[(size_t)e] > 0.5;
              REQUIRE(blmm_multi_last_lod_threshold(mc, 0.5, cap, ti.data(), tj.data(), tl.data(), &cnt) == BLMM_OK);
              REQUIRE(cnt == cref);
              for (int64_t e = 0; e < (cnt < cap ? cnt : cap); ++e) REQUIRE(tl[(size_t)e] == Lref[(size_t)(tj[(size_t)e] * p + ti[(size_t)e])] && tl[(size_t)e] > 0.5);
            }
            if (gather != BLMM_GATHER_HOST_SHARDS && m > 0) {
              double* dL = nullptr; double* dh = nullptr; int64_t ld = 0, lo = 0, hi = 0;
              REQUIRE(blmm_multi_device_result(mc, R - 1, &dL, &ld, &lo, &hi, &dh) == BLMM_OK);
              REQUIRE(ld == p && hi <= m && lo <= hi);
            }
          }
        }
      }
    }
    blmm_destroy_multi(mc);
    (void)hipGetDevice(&cur); REQUIRE(cur == cfg);
  }
}

static void test_readers(const char* dir) {
  const std::string base(dir);
  auto write = [&](const char* name, const std::string& txt) { const std::string f = base + "/" + name; FILE* fp = std::fopen(f.c_str(), "wb"); REQUIRE(fp); std::fwrite(txt.data(), 1, txt.size(), fp); std::fclose(fp); return f; };
  blmm_table* t = nullptr;
  const std::string good = write("good.csv", "\"id\",\"a\",\"b\",\"c\"\r\n\"x,1\",1.5,2.5,3.5\r\n\"y\",4,5,6\n");
  REQUIRE(blmm_read_csv(good.c_str(), 1, 1, 1, 0, &t) == BLMM_OK);
  REQUIRE(blmm_table_rows(t) == 2 && blmm_table_cols(t) == 3);
  std::vector<double> v(6);
  REQUIRE(blmm_table_copy(t, v.data()) == BLMM_OK);
  REQUIRE(v[0] == 1.5 && v[1] == 4.0 && v[5] == 6.0);
  blmm_table_free(t);
  REQUIRE(blmm_read_csv(good.c_str(), 1, 1, 2, 1, &t) == BLMM_OK);       // every other column, last dropped
  REQUIRE(blmm_table_cols(t) == 1);
  blmm_table_free(t);
  for (const char* bad : {"id,a\nx,1\ny,2,3\n", "id,a\nx,abc\n", "", "id,a\n", "id,a\nx,\n", "id\n\"unterminated,1\n"}) {
    const std::string f = write("bad.csv", bad);
    t = nullptr;
    const int rc = blmm_read_csv(f.c_str(), 1, 1, 1, 0, &t);
    if (rc == BLMM_OK) blmm_table_free(t);                                // whatever it decides, it must not read out of bounds
  }
  REQUIRE(blmm_read_csv((base + "/missing.csv").c_str(), 1, 1, 1, 0, &t) != BLMM_OK);
  // Helium: 56-byte header (nrow, ncol, ...) + column-major float64; truncated payloads must be rejected
  std::string he(56, '\0');
  const int64_t nr = 3, nc = 2;
  std::memcpy(&he[0], &nr, 8); std::memcpy(&he[8], &nc, 8);
  std::string payload(48, '\0');
  for (int i = 0; i < 6; ++i) { const double d = i + 0.25; std::memcpy(&payload[8 * i], &d, 8); }
  const std::string hf = write("m.he", he + payload);
  REQUIRE(blmm_read_he(hf.c_str(), &t) == BLMM_OK);
  REQUIRE(blmm_table_rows(t) == 3 && blmm_table_cols(t) == 2);
  blmm_table_free(t);
  const std::string hs = write("short.he", he + payload.substr(0, 40));
  REQUIRE(blmm_read_he(hs.c_str(), &t) != BLMM_OK);
  const std::string hh = write("hdr.he", he.substr(0, 20));
  REQUIRE(blmm_read_he(hh.c_str(), &t) != BLMM_OK);
}

int main(int argc, char** argv) {
  test_copy_pool();
  test_multi();
  test_readers(argc > 1 ? argv[1] : "/tmp");
  std::printf("sanitize: copy pool, multi-device workers, readers ok\n");
  return 0;
}
