// Device side of the peer-memory exchange (peer_comm.hip): buffer layout, the descriptor a kernel gets, and the
// per-channel exchange that the batch-norm finalize kernels (pointwise.hip) run in place of a separate all-reduce launch.
#pragma once
#include "common.hpp"

// Ring of slots, collective i in slot i % NSLOT.  On ONE stream four would do (a rank can contribute to collective i + 1 only
// after it has read collective i, and finish i + 1 only once every rank has contributed to it, i.e. read i).  With the
// branch networks on streams of their own the collectives of a step run in any order, and the ring must hold a whole
// step: a rank can only start step t + 1 after every rank has contributed to every collective of step t, and its own
// backward collectives only after every rank has contributed to its forward ones -- so a slot written for collective j
// last served collective j - NSLOT of a step every rank has left, as long as a step issues fewer than NSLOT / 2
// collectives (the CVAE: 58; dist.Sync checks it).
constexpr int PC_NSLOT = 256;
constexpr int PC_MAXN = 1024;            // doubles per contribution
constexpr int PC_MAXW = 16;              // ranks

// buffer layout: data [NSLOT][world][MAXN] doubles; message flags [NSLOT][world] 64-bit words (one 128-byte line each);
// element flags [NSLOT][world][MAXN] 64-bit words (the per-channel exchange: one flag per channel)
__host__ __device__ inline size_t pc_data_off(int world, int slot, int r) { return ((size_t)slot * world + r) * PC_MAXN * sizeof(double); }
__host__ __device__ inline size_t pc_flag_off(int world, int slot, int r) {
  return (size_t)PC_NSLOT * world * PC_MAXN * sizeof(double) + ((size_t)slot * world + r) * 128;
}
__host__ __device__ inline size_t pc_eflag_off(int world, int slot, int r) {
  return pc_flag_off(world, PC_NSLOT, 0) + ((size_t)slot * world + r) * PC_MAXN * sizeof(unsigned long long);
}
inline size_t pc_bytes(int world) { return pc_eflag_off(world, PC_NSLOT, 0); }

struct PeerDev {
  char* peer[PC_MAXW];                   // every rank's buffer in this address space
  int rank, world;                       // world == 0: no exchange
  unsigned long long seq;                // this collective's number (1-based, the same on every rank)
  unsigned long long* status;            // timeouts seen by this rank's kernels
  long long spin_limit;
};

// host side (peer_comm.hip): the next collective of the communicator bound to this thread (bp_peer_bind); false: none bound
bool bp_peer_next(PeerDev* out);

// ONE thread: v[k] (element e + k * stride of the contribution, k < NV) summed over the ranks in rank order -- the same
// order on every rank.  The protocol of peer_all_reduce_kernel per element: write-through system-scope stores of the
// payload into every rank's buffer, their acknowledgement awaited, then the element's flag; relaxed polls of the
// flags in the own buffer; bounded spins (a timeout is recorded in the status word, the values are then local only).
template <int NV>
__device__ inline bool peer_exchange(const PeerDev& pd, int e, int stride, double (&v)[NV]) {
  const int slot = (int)(pd.seq % PC_NSLOT);
  for (int p = 0; p < pd.world; ++p) {
    double* dst = reinterpret_cast<double*>(pd.peer[p] + pc_data_off(pd.world, slot, pd.rank));
#pragma unroll
    for (int k = 0; k < NV; ++k) __hip_atomic_store(dst + e + k * stride, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int p = 0; p < pd.world; ++p) {
    unsigned long long* f = reinterpret_cast<unsigned long long*>(pd.peer[p] + pc_eflag_off(pd.world, slot, pd.rank)) + e;
    __hip_atomic_store(f, pd.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  for (int r = 0; r < pd.world; ++r) {
    const unsigned long long* f = reinterpret_cast<const unsigned long long*>(pd.peer[pd.rank] + pc_eflag_off(pd.world, slot, r)) + e;
    long long spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != pd.seq) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > pd.spin_limit) { atomicAdd(pd.status, 1ull); return false; }
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double s = 0.0;
    for (int r = 0; r < pd.world; ++r) {
      const double* src = reinterpret_cast<const double*>(pd.peer[pd.rank] + pc_data_off(pd.world, slot, r)) + e + k * stride;
      s += __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    v[k] = s;
  }
  return true;
}
