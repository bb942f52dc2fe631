// ThreadSanitizer driver for the C-ABI host shim (tools/tsan_host_shim.sh): N threads call into the library at the same time -- the
// situation of the training step, where autograd worker threads run the backward entry points while the main thread runs forward
// ones.  No GPU is needed: calls either fail validation or fail at the launch; what is exercised is the host code in between (the
// thread-local error text, the per-device LDS opt-ins, the knob statics, the tile-counter ring of the persistent GEMM).
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

typedef int (*gemm_fn)(const void*, const void*, void*, int64_t, int, int, int64_t, int64_t, int64_t, int, const void*, int64_t, void*);
typedef int (*attn_fn)(const void*, const uint64_t*, void*, float*, int, int, int, int, int, int64_t, int64_t, float, void*);
typedef int (*pool_fn)(const void*, const int64_t*, const int32_t*, float*, float*, int, int, int, int, int, void*);
typedef int (*rms_bwd_fn)(const void*, const void*, const void*, const void*, void*, float*, float*, int64_t, int, float, void*);
typedef int (*rms_fn)(const void*, const void*, void*, int64_t, int, float, void*);
typedef const char* (*err_fn)(void);
typedef int64_t (*ws_fn)(int, int, int, int);

int main(int argc, char** argv) {
  void* h = dlopen(argc > 1 ? argv[1] : "tools/_tsan/libgritlm_hip_tsan.so", RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  gemm_fn gemm = (gemm_fn)dlsym(h, "grit_gemm_bf16_nt");
  attn_fn attn = (attn_fn)dlsym(h, "grit_attn_bidir_fwd");
  pool_fn pool = (pool_fn)dlsym(h, "grit_pool_norm_fwd");
  rms_bwd_fn rms_bwd = (rms_bwd_fn)dlsym(h, "grit_rmsnorm_bwd");
  rms_fn rms = (rms_fn)dlsym(h, "grit_rmsnorm_fwd");
  err_fn err = (err_fn)dlsym(h, "grit_last_error_string");
  ws_fn ws = (ws_fn)dlsym(h, "grit_attn_decode_workspace_floats");
  if (!gemm || !attn || !pool || !rms_bwd || !rms || !err || !ws) { fprintf(stderr, "missing symbol\n"); return 2; }
  static char buf[1 << 16] __attribute__((aligned(256)));
  const int n_threads = 8, iters = 200;
  std::vector<std::thread> th;
  std::vector<long> bad(n_threads, 0);
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([&, t] {
      for (int i = 0; i < iters; ++i) {
        // valid-shaped calls (fail at the launch: no device) and invalid ones (fail validation), interleaved per thread
        int r1 = gemm(buf, buf, buf, 4096, 4096, 4096, 4096, 4096, 4096, (t + i) & 1, buf, 4096, nullptr);
        int r2 = gemm(nullptr, buf, buf, 4096, 4096, 4096, 4096, 4096, 4096, 0, nullptr, 0, nullptr);
        int r3 = attn(buf, (const uint64_t*)buf, buf, (float*)buf, 4, 512, 32, 8, 128, 6144, 4096, 0.088f, nullptr);
        int r4 = attn(buf, (const uint64_t*)buf, buf, (float*)buf, 4, 512, 32, 8, 64 + t, 6144, 4096, 0.088f, nullptr);
        int r5 = pool(buf, (const int64_t*)buf, nullptr, (float*)buf, (float*)buf, 4, 512, 4096, i & 3, 1, nullptr);
        int r6 = rms_bwd(buf, buf, buf, nullptr, buf, (float*)buf, (float*)buf, 2048, 4096, 1e-5f, nullptr);
        int r7 = rms(buf, buf, buf, 2048, 4100 + 8 * t, 1e-5f, nullptr);     // H not a multiple of 8: rejected, with a per-thread message
        const char* e = err();
        char want[32]; snprintf(want, sizeof want, "%d", 4100 + 8 * t);
        if (r7 < 0 && !strstr(e, want)) bad[t]++;                                    // another thread's message leaked into this one
        if (r2 != -1 || r4 != -2 || r1 >= 0 || r3 >= 0 || r5 >= 0 || r6 >= 0) bad[t] += 1000;
        (void)ws(1 + t, 32, 8, 2048);
      }
    });
  for (auto& x : th) x.join();
  long total = 0;
  for (long b : bad) total += b;
  printf("threads %d x %d iterations; unexpected results %ld\n", n_threads, iters, total);
  return total ? 1 : 0;
}
