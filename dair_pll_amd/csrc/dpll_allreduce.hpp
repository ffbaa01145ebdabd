// dpll_allreduce.hpp -- one-shot all-reduce of the tiny [loss | gradients] vector over xGMI peer memory.
//
// The path's only collective is a SUM of 16 (cube) / 30 (elbow) numbers per step: pure latency.  A ring or
// tree collective pays several hops plus a kernel of its own; here every rank pushes its vector straight into
// every peer's receive buffer (one xGMI store hop), then sums what arrived in rank order, so all ranks get the
// bitwise-identical result in a single ~2 us kernel that lives in the step's hipGraph.
//
// Protocol (per call s = 1, 2, ...; s is kept in device memory so a replayed graph advances it):
//   * receive buffer of rank r: granules[2][world][kMaxWords], a granule = one naturally aligned 8-byte
//     {32 data bits, tag = s} written by ONE system-scope store -- the data is its own flag, no fences;
//   * rank r stores its words into slot s & 1, row r of every rank's buffer (its own included);
//   * rank r then polls its own buffer until rows 0..world-1 of slot s & 1 all carry tag s and sums them;
//   * two slots suffice: a rank can enter call s + 2 (same slot as s) only after receiving every peer's
//     call-(s+1) data, which a peer sends only after it finished reading slot s & 1 of call s.
// Buffers are device memory allocated UNCACHED (fine-grained) and shared through hipIpc handles, so polls and
// remote stores are never served from a stale L2 line.  Every spin is bounded; a timeout raises an error word
// that the host checks (dpll_ar_status), and the words that never arrived are replaced by NaN so that the reduced row
// cannot be mistaken for a result -- the caller then falls back to RCCL.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dpll_arx {

constexpr int kMaxWorld = 16;
constexpr int kMaxWords = 256;  // 32-bit words per call (128 doubles / 256 floats)
constexpr unsigned long long kSpinLimit = 5000000ull;  // s_memrealtime ticks (100 MHz): 50 ms, then give up
constexpr uint32_t kPoison = 0x7ff80000u;              // a word that reads as NaN both as a float and as the high half of a double

struct Peers {
  unsigned long long* recv[kMaxWorld];  // receive buffers of all ranks as seen from this process
};

__device__ __forceinline__ void store_granule(unsigned long long* p, uint32_t word, uint32_t tag) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)word, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long load_granule(unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The exchange itself, for one workgroup: `words` (LDS, n_words 32-bit words of this rank) go to every rank's
// receive buffer, then `gathered[from][i]` (LDS) is filled with every rank's words of call `seq`.  Ends with a
// barrier.  Shared by the stand-alone kernel and by the finalize kernel that ends a loss launch (fused form).
__device__ __forceinline__ void exchange_words(const uint32_t* words, uint32_t (*gathered)[kMaxWords], int n_words,
                                               const Peers& peers, int rank, int world, uint32_t seq, uint32_t* err) {
  const int slot = (int)(seq & 1u);
  // push: one 8-byte store per (peer, word)
  for (int idx = threadIdx.x; idx < world * n_words; idx += blockDim.x) {
    const int peer = idx / n_words, i = idx % n_words;
    store_granule(peers.recv[peer] + ((size_t)(slot * world + rank) * kMaxWords + i), words[i], seq);
  }
  // pull: wait for every rank's row in my own buffer
  unsigned long long* mine = peers.recv[rank];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  bool timed_out = false;
  for (int idx = threadIdx.x; idx < world * n_words; idx += blockDim.x) {
    const int from = idx / n_words, i = idx % n_words;
    unsigned long long* p = mine + ((size_t)(slot * world + from) * kMaxWords + i);
    unsigned long long g = load_granule(p);
    while ((uint32_t)(g >> 32) != seq) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > kSpinLimit) { timed_out = true; break; }
      __builtin_amdgcn_s_sleep(1);
      g = load_granule(p);
    }
    // a peer that never arrived must not contribute a stale or partial value silently: the sum becomes NaN
    gathered[from][i] = timed_out ? kPoison : (uint32_t)g;
  }
  if (timed_out) atomicExch(err, 1u);
  __syncthreads();
}

// element e of the exchanged vectors summed in rank order: every rank computes bitwise the same value
template <typename T> __device__ __forceinline__ T sum_over_ranks(uint32_t (*gathered)[kMaxWords], int world, int e) {
  constexpr int kWordsPer = sizeof(T) / 4;
  double s = 0.0;
  for (int from = 0; from < world; ++from) {
    T v;
    __builtin_memcpy(&v, &gathered[from][e * kWordsPer], sizeof(T));
    s += double(v);
  }
  return T(s);
}

// data: n elements of T (float or double), summed in place over the ranks.  One workgroup of 256 threads.
template <typename T>
__global__ __launch_bounds__(256) void allreduce_kernel(T* __restrict__ data, int n, Peers peers, int rank, int world,
                                                        uint32_t* __restrict__ seq_ptr, uint32_t* __restrict__ err) {
  constexpr int kWordsPer = sizeof(T) / 4;
  __shared__ uint32_t words[kMaxWords];
  __shared__ uint32_t gathered[kMaxWorld][kMaxWords];
  const int n_words = n * kWordsPer;
  const uint32_t seq = *seq_ptr + 1u;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(data);
  for (int i = threadIdx.x; i < n_words; i += blockDim.x) words[i] = src[i];
  __syncthreads();
  exchange_words(words, gathered, n_words, peers, rank, world, seq, err);
  for (int e = threadIdx.x; e < n; e += blockDim.x) data[e] = sum_over_ranks<T>(gathered, world, e);
  if (threadIdx.x == 0) *seq_ptr = seq;
}

}  // namespace dpll_arx
