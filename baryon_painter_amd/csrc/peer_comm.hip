// Small-message all-reduce over peer memory (xGMI between the GPUs of a node), without RCCL.
//
// Why: data-parallel training with the reference's arithmetic (global-batch BatchNorm2d,
// /root/reference/baryon_painter/painter.py:224 feeds the WHOLE minibatch through nn.BatchNorm2d) needs the per-channel
// batch-norm sums of every layer, forward and backward, summed over the ranks before the next kernel can run: 44
// dependent collectives of <= 2 KB per step.  Through torch.distributed each costs 16-18 us before a byte moves
// (profiles/r03_rccl_one_rank.txt: event hand-off to the communicator's stream and back, an RCCL launch), 0.76 ms per step
// = 7.7 % of the bf16 step at world size 1.  Here ONE 256-thread kernel on the caller's stream does the whole exchange:
//   * every rank owns a buffer of fine-grained device memory (hipExtMallocWithFlags: coherent at system scope) mapped
//     into every other rank's address space through hipIpc handles (exchanged once, by the host, over the process group);
//   * contribution: each rank stores its n doubles into slot [seq % NSLOT][rank] of EVERY rank's buffer (write-through
//     system-scope stores, straight over xGMI), waits for their acknowledgement, then stores the sequence number into
//     that slot's flag (no cache write-back / invalidate anywhere: every byte of the exchange bypasses the caches);
//   * it polls the flags of all ranks in ITS OWN buffer (system-scope loads of local memory), then sums the world_size
//     contributions in rank order -- the same order on every rank: results are bitwise identical across ranks, as an
//     all-reduce must be for the replicas to stay in step;
//   * slot reuse: a ring of NSLOT slots that holds more than a whole step's collectives (peer_dev.hpp);
//   * every spin is bounded: on a timeout the kernel records it in the communicator's status word and returns (the step's
//     numbers are then wrong; the host checks the word after the step and raises).
// The host side (baryon_painter_amd/dist.py) validates the path at start-up with known data on every rank and falls back
// to RCCL for the whole run if any rank saw a wrong sum or a timeout.
//
// Fused form (bp_peer_bind): while a communicator is bound to the calling thread, the kernels that finish a layer's
// batch-norm statistics (pointwise.hip: sum_partials_bn_kernel, sum_partials_bnbwd_kernel -- one workgroup per channel)
// exchange their channel's sums themselves (peer_dev.hpp: peer_exchange, one flag per channel) between summing the
// partial rows and the finalize arithmetic: a data-parallel step then has the launch count of the single-device step.
#include "peer_dev.hpp"
#include <cstring>
#include <new>

namespace {

struct PeerComm {
  int rank, world;
  unsigned long long seq;                // collectives issued so far (host side; every rank counts alike)
  char* local;                           // this rank's buffer
  char* peer[PC_MAXW];                   // every rank's buffer in this address space (peer[rank] == local)
  bool opened[PC_MAXW];
  unsigned long long* status;            // device word: number of timeouts seen by this rank's kernels
  hipIpcMemHandle_t handle;
};

thread_local PeerComm* t_bound = nullptr;            // bp_peer_bind

struct PcArgs {
  char* peer[PC_MAXW];
  int rank, world, n;
  unsigned long long seq;                // 1-based
  double* data;
  unsigned long long* status;
  long long spin_limit;
};

__global__ __launch_bounds__(256) void peer_all_reduce_kernel(PcArgs a) {
  const int slot = (int)(a.seq % PC_NSLOT);
  const int tid = threadIdx.x;
  // ---- contribute: my n doubles into slot [slot][rank] of every rank's buffer
  for (int i = tid; i < a.n * a.world; i += 256) {
    const int p = i / a.n, j = i - p * a.n;
    double* dst = reinterpret_cast<double*>(a.peer[p] + pc_data_off(a.world, slot, a.rank)) + j;
    __hip_atomic_store(dst, a.data[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // The payload before the flags WITHOUT a release fence: a system-scope release is buffer_wbl2 sc0 sc1 -- a write-back of
  // every dirty line of this XCD's L2, i.e. of the convolution output the previous kernel just wrote (measured: 9.4 us per
  // collective with the fence pair, most of it the write-back).  The payload stores are write-through system-scope stores
  // (sc0 sc1: they leave for the owner's memory at once); every storing wave waits for their acknowledgement (vmcnt(0)),
  // the workgroup meets, and only then are the flags stored -- to the same memories, strictly later.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < a.world) {
    unsigned long long* f = reinterpret_cast<unsigned long long*>(a.peer[tid] + pc_flag_off(a.world, slot, a.rank));
    __hip_atomic_store(f, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ---- wait for every rank's flag in MY buffer
  __shared__ int timed_out;
  if (tid == 0) timed_out = 0;
  __syncthreads();
  if (tid < a.world) {
    const unsigned long long* f = reinterpret_cast<const unsigned long long*>(a.peer[a.rank] + pc_flag_off(a.world, slot, tid));
    long long spins = 0;
    // (relaxed polls: an acquire load would invalidate the caches on every iteration; the payload is read below with
    //  system-scope loads that bypass them, issued after the poll has matched and the workgroup has met)
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > a.spin_limit) { timed_out = 1; break; }
    }
  }
  __syncthreads();
  if (timed_out) {
    if (tid == 0) atomicAdd(a.status, 1ull);
    return;
  }
  // ---- sum in rank order (system-scope loads: the lines were written by other devices)
  for (int j = tid; j < a.n; j += 256) {
    double s = 0.0;
    for (int r = 0; r < a.world; ++r) {
      const double* src = reinterpret_cast<const double*>(a.peer[a.rank] + pc_data_off(a.world, slot, r)) + j;
      s += __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    a.data[j] = s;
  }
}

// the per-channel exchange on its own (start-up self-test of the fused path): workgroup ch sums {data[ch], data[c + ch]} over the ranks
__global__ __launch_bounds__(64) void peer_exchange_check_kernel(PeerDev pd, double* data, int c) {
  if (threadIdx.x != 0) return;
  const int ch = blockIdx.x;
  double v[2] = {data[ch], data[c + ch]};
  peer_exchange<2>(pd, ch, c, v);
  data[ch] = v[0]; data[c + ch] = v[1];
}

}  // namespace

bool bp_peer_next(PeerDev* out) {
  PeerComm* c = t_bound;
  if (!c) { out->world = 0; return false; }
  for (int r = 0; r < PC_MAXW; ++r) out->peer[r] = r < c->world ? c->peer[r] : nullptr;
  out->rank = c->rank; out->world = c->world; out->seq = ++c->seq; out->status = c->status;
  out->spin_limit = 1ll << 22;
  return true;
}

extern "C" {

int bp_peer_bind(void* comm) {
  PeerComm* c = reinterpret_cast<PeerComm*>(comm);
  if (c)
    for (int r = 0; r < c->world; ++r)
      if (!c->peer[r]) return BP_EINVAL;
  t_bound = c;
  return BP_OK;
}

int bp_peer_exchange_check(void* comm, double* data, int c, void* stream) {
  PeerComm* cm = reinterpret_cast<PeerComm*>(comm);
  if (!cm || !data || c < 1 || 2 * c > PC_MAXN) return BP_EINVAL;
  PeerComm* keep = t_bound;
  t_bound = cm;
  PeerDev pd;
  (void)bp_peer_next(&pd);
  t_bound = keep;
  hipLaunchKernelGGL(peer_exchange_check_kernel, dim3(c), dim3(64), 0, bp_stream(stream), pd, data, c);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_peer_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }
int bp_peer_max_doubles(void) { return PC_MAXN; }
int bp_peer_slots(void) { return PC_NSLOT; }

int bp_peer_create(int rank, int world, void** comm_out, void* handle_out) {
  if (!comm_out || !handle_out || world < 1 || world > PC_MAXW || rank < 0 || rank >= world) return BP_EINVAL;
  PeerComm* c = new (std::nothrow) PeerComm();
  if (!c) return BP_EINVAL;
  c->rank = rank; c->world = world; c->seq = 0;
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, pc_bytes(world) + 256, hipDeviceMallocFinegrained) != hipSuccess) { delete c; return BP_EUNSUPPORTED; }
  c->local = reinterpret_cast<char*>(p);
  if (hipMemset(p, 0, pc_bytes(world) + 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); delete c; return BP_ELAUNCH; }
  c->status = reinterpret_cast<unsigned long long*>(c->local + pc_bytes(world));
  for (int r = 0; r < PC_MAXW; ++r) { c->peer[r] = nullptr; c->opened[r] = false; }
  c->peer[rank] = c->local;
  if (hipIpcGetMemHandle(&c->handle, p) != hipSuccess) {
    (void)hipGetLastError();
    if (world > 1) { (void)hipFree(p); delete c; return BP_EUNSUPPORTED; }
    memset(&c->handle, 0, sizeof(c->handle));               // (a single rank needs no handle)
  }
  memcpy(handle_out, &c->handle, sizeof(c->handle));
  *comm_out = c;
  return BP_OK;
}

int bp_peer_open(void* comm, const void* handles) {
  PeerComm* c = reinterpret_cast<PeerComm*>(comm);
  if (!c || !handles) return BP_EINVAL;
  const hipIpcMemHandle_t* h = reinterpret_cast<const hipIpcMemHandle_t*>(handles);
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return BP_EUNSUPPORTED; }
    c->peer[r] = reinterpret_cast<char*>(p);
    c->opened[r] = true;
  }
  return BP_OK;
}

int bp_peer_all_reduce(void* comm, double* data, int n, int64_t spin_limit, void* stream) {
  PeerComm* c = reinterpret_cast<PeerComm*>(comm);
  if (!c || !data || n < 1 || n > PC_MAXN) return BP_EINVAL;
  PcArgs a{};
  for (int r = 0; r < c->world; ++r) {
    if (!c->peer[r]) return BP_EINVAL;
    a.peer[r] = c->peer[r];
  }
  a.rank = c->rank; a.world = c->world; a.n = n; a.seq = ++c->seq; a.data = data; a.status = c->status;
  a.spin_limit = spin_limit > 0 ? spin_limit : (1ll << 22);
  hipLaunchKernelGGL(peer_all_reduce_kernel, dim3(1), dim3(256), 0, bp_stream(stream), a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// number of timeouts recorded so far (synchronises with the device)
int64_t bp_peer_status(void* comm) {
  PeerComm* c = reinterpret_cast<PeerComm*>(comm);
  if (!c) return -1;
  unsigned long long v = 0;
  if (hipMemcpy(&v, c->status, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int64_t)v;
}

int bp_peer_destroy(void* comm) {
  PeerComm* c = reinterpret_cast<PeerComm*>(comm);
  if (!c) return BP_EINVAL;
  for (int r = 0; r < c->world; ++r)
    if (c->opened[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
  (void)hipFree(c->local);
  delete c;
  return BP_OK;
}

}  // extern "C"
