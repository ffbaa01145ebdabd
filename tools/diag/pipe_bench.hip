// Diagnostic: the pipelined ICNN GEMM kernels (csrc/dpll_icnn_pipe.hip) alone, on random operands, with shader-clock stamps
// of every workgroup: where a launch's time goes (prologue, each chain of the first three tiles, drain).  Timing only --
// correctness is the business of tests/test_hip_mesh.py.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -DDPLL_PIPE_STAMPS -I dair_pll_amd/csrc \
//         -o tools/diag/pipe_bench tools/diag/pipe_bench.hip && tools/diag/pipe_bench [batch]
#include "../../dair_pll_amd/csrc/dpll_icnn_pipe.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const long long batch = argc > 1 ? atoll(argv[1]) : 4096, N = 4 * batch;
  const long long tiles = dpll_pipe::tiles(N);
  std::vector<float> h(256 * 256), hx(13 * batch), hw(3 * 256);
  srand(1);
  for (auto& v : h) v = 0.002f * (rand() / (float)RAND_MAX);
  for (auto& v : hw) v = (rand() / (float)RAND_MAX) - 0.5f;
  for (long long i = 0; i < batch; ++i) {
    float q[4], n = 0;
    for (int j = 0; j < 4; ++j) { q[j] = rand() / (float)RAND_MAX - 0.5f; n += q[j] * q[j]; }
    for (int j = 0; j < 4; ++j) hx[13 * i + j] = q[j] / sqrtf(n);
    for (int j = 4; j < 13; ++j) hx[13 * i + j] = 0.1f;
  }
  float *F, *x, *Wd0, *Wd1, *wout, *pert, *a, *U0, *P, *RB, *Vb, *slabs;
  uint32_t* M1; double* partial;
  hipMalloc(&F, 4 * 256 * 256); hipMalloc(&x, 4 * 13 * batch); hipMalloc(&Wd0, 4 * 768); hipMalloc(&Wd1, 4 * 768);
  hipMalloc(&wout, 4 * 256); hipMalloc(&pert, 4 * 12); hipMalloc(&a, 4 * 256); hipMalloc(&U0, 4 * 8192 * (tiles + 1));
  hipMalloc(&P, 4 * 3 * N); hipMalloc(&RB, 4 * 3 * N); hipMalloc(&Vb, 4 * 8192 * (tiles + 1)); hipMalloc(&M1, 4 * 8 * N);
  hipMalloc(&partial, 8 * 7 * 256 * 256); hipMalloc(&slabs, 4 * 65536 * 64);
  hipMemcpy(F, h.data(), 4 * 65536, hipMemcpyHostToDevice); hipMemcpy(x, hx.data(), 4 * 13 * batch, hipMemcpyHostToDevice);
  hipMemcpy(Wd0, hw.data(), 4 * 768, hipMemcpyHostToDevice); hipMemcpy(Wd1, hw.data(), 4 * 768, hipMemcpyHostToDevice);
  hipMemcpy(wout, h.data(), 4 * 256, hipMemcpyHostToDevice); hipMemcpy(a, h.data(), 4 * 256, hipMemcpyHostToDevice);
  hipMemset(pert, 0, 48); hipMemset(RB, 0, 4 * 3 * N); hipMemset(M1, 0x5a, 4 * 8 * N); hipMemset(U0, 0, 4 * 8192 * (tiles + 1));
  dpll::IcnnWeights<float> w{F, Wd0, Wd1, wout, pert};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[6] = {"fwd1", "fwd2", "bwd1", "fwd1 bf16x2", "fwd2 bf16x2", "bwd1 bf16x2"};
  for (int k = 0; k < 6; ++k) {
    auto launch = [&]() {
      if (k == 0) dpll_pipe::fwd1(0, x, 13, N, w, F, M1);
      if (k == 1) dpll_pipe::fwd2(0, x, 13, N, w, F, a, M1, U0, P);
      if (k == 2) dpll_pipe::bwd1(0, x, 13, N, w, F, a, M1, U0, RB, partial, Vb);
      if (k == 3) dpll_pipe::fwd1_bf16(0, x, 13, N, w, F, M1);   // (F read as two bf16 planes: 256 KB of arbitrary finite numbers)
      if (k == 4) dpll_pipe::fwd2_bf16(0, x, 13, N, w, F, a, M1, U0, P);
      if (k == 5) dpll_pipe::bwd1_bf16(0, x, 13, N, w, F, a, M1, U0, RB, partial, Vb);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
#ifdef DPLL_PIPE_STAMPS
    static unsigned long long st[256][16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_pipe_stamps), sizeof(st));
    const int nb = dpll_pipe::blocks(N);
    printf("%s batch %lld: %.2f us per launch (back to back); stamps in units of 100 s_memtime ticks (the counter runs at the shader clock, ~2.1 GHz under this load), median over %d workgroups, relative to entry:\n", names[k], batch, ms * 1e3 / reps, nb);
    const char* lab[16] = {"entry", "loop top", "t0 start", "t0 chain0", "t0 chain1", "t0 commit", "t1 start", "t1 chain0", "t1 chain1", "t1 commit",
                           "loads issued", "ring zeroed", "rows in ring", "barrier", "drained", "end"};
    for (int i : {10, 11, 12, 13, 1, 2, 3, 4, 5, 6, 7, 8, 9, 14, 15}) {
      std::vector<long long> d;
      for (int b = 0; b < nb; ++b) if (st[b][i] > st[b][0]) d.push_back((long long)(st[b][i] - st[b][0]));
      if (d.empty()) continue;
      std::sort(d.begin(), d.end());
      printf("  %-10s %8.2f (max %.2f)\n", lab[i], d[d.size() / 2] * 0.01, d.back() * 0.01);
    }
    unsigned long long lo = ~0ull, hi = 0;
    for (int b = 0; b < nb; ++b) { lo = std::min(lo, st[b][0]); hi = std::max(hi, st[b][15]); }
    printf("  first entry -> last end over the grid: %.2f us\n", (hi - lo) * 0.01);
#else
    printf("%s batch %lld: %.2f us per launch (back to back)\n", names[k], batch, ms * 1e3 / reps);
#endif
  }
  return 0;
}
