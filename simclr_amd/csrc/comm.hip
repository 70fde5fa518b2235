// One-shot exchange of the SyncBatchNormalization statistics over peer-mapped memory (collective C of SURVEY section 2.3:
// /root/reference/tf2/resnet.py:50-60 -- tf.keras.layers.experimental.SyncBatchNormalization all-reduces the per-replica
// [2, C] moment sums of every BatchNorm, ~112 times per ResNet-50 training step, each on the critical path).
//
// A collective library pays its full protocol (ring / tree steps, proxy wake-ups, ~20 us at 8 ranks) for every one of
// these <= 16 KB messages.  Here every rank owns a MAILBOX that all its peers map (hipIpc): [2 generations][R slots] of
// payload + a sequence flag.  One launch of one workgroup per exchange:
//   1. write my fp64 block into slot `rank` of EVERY peer's mailbox (8-byte system-scope stores over xGMI),
//   2. __threadfence_system(), then store the sequence number into that slot's flag (system scope),
//   3. spin (bounded) until the R flags of MY mailbox show this sequence number, acquire,
//   4. out = sum of the R slots in RANK ORDER -- every rank adds the same numbers in the same order, so all replicas get
//      bit-identical statistics and the run-to-run determinism of the step is kept.
// Two generations (seq & 1): a rank can be at most ONE exchange ahead of the slowest peer -- it cannot finish exchange
// s + 1 before every peer has published s + 1, which a peer does only after it has read generation s.
// The mailbox is allocated by the library (uncached device memory where the runtime offers it: flags and payload are then
// never served from a stale L2 line) and exported / imported as hipIpc handles; exchanging the 64-byte handles between
// the processes is the caller's job (simclr_amd/comm.py uses torch.distributed.all_gather_object).
// RCCL stays the fallback and the transport of collectives A and B.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int kFlagBytes = 64;      // one cache line per flag

struct CommP {
  unsigned char* peer[16];          // mailbox of every rank as mapped in THIS process (peer[rank] = the local one)
  int rank, world;
  int max_doubles;                  // payload capacity of a slot
  unsigned seq;
  long long timeout_ticks;          // bound of the arrival wait in wall_clock64() ticks (constant-rate counter, 100 MHz on gfx9)
};

__device__ __forceinline__ size_t slot_bytes(int max_doubles) { return (size_t)max_doubles * 8 + kFlagBytes; }

__global__ __launch_bounds__(256) void stats_exchange(const CommP p, const double* __restrict__ in, double* __restrict__ out,
                                                     int count, int* __restrict__ status) {
  const int tid = threadIdx.x;
  const int gen = (int)(p.seq & 1u);
  const size_t sb = slot_bytes(p.max_doubles);
  // 1. my block -> slot `rank` of every mailbox
  for (int r = 0; r < p.world; ++r) {
    double* dst = (double*)(p.peer[r] + ((size_t)gen * p.world + p.rank) * sb + kFlagBytes);
    for (int i = tid; i < count; i += 256)
      __hip_atomic_store(dst + i, in[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 2. publish: every thread's stores are ordered before the flags by a system-scope fence + the barrier
  __threadfence_system();
  __syncthreads();
  if (tid < p.world) {
    unsigned* flag = (unsigned*)(p.peer[tid] + ((size_t)gen * p.world + p.rank) * sb);
    __hip_atomic_store(flag, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 3. wait for the R blocks addressed to me (bounded: a dead peer must not hang the GPU; status reports it)
  __shared__ int bad;
  if (tid == 0) bad = 0;
  __syncthreads();
  if (tid < p.world) {
    const unsigned* flag = (const unsigned*)(p.peer[p.rank] + ((size_t)gen * p.world + tid) * sb);
    // Bounded by WALL CLOCK (an iteration count is clock- and contention-dependent), generously: a rank may lag by an
    // input-pipeline stall, a checkpoint write or a first-step lazy load, all of which a collective library would simply wait out.
    // Default 600 s -- the order of the collective library's own watchdog; SIMCLR_PEER_STATS_TIMEOUT_S overrides.
    const long long t0 = wall_clock64();
    int polls = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != p.seq) {
      __builtin_amdgcn_s_sleep(4);
      if ((++polls & 255) == 0 && wall_clock64() - t0 > p.timeout_ticks) { atomicAdd(&bad, 1); break; }
    }
  }
  __threadfence_system();
  __syncthreads();
  // STICKY: the word only ever grows (number of peer arrivals missed since the mailbox was created); a healthy exchange never
  // clears what an earlier one reported, so a host that looks once per step cannot miss a time-out
  const int missed = bad;
  if (tid == 0 && status && missed) atomicAdd(status, missed);
  // 4. fixed-order sum.  After a time-out the slots hold stale or partial data: the result is POISONED with NaN instead, so the
  // statistics -- and with them the loss -- of this replica fail loudly rather than drift apart from the other replicas'
  for (int i = tid; i < count; i += 256) {
    double s = 0.0;
    for (int r = 0; r < p.world; ++r) {
      const double* src = (const double*)(p.peer[p.rank] + ((size_t)gen * p.world + r) * sb + kFlagBytes);
      s += __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    out[i] = missed ? __builtin_nan("") : s;
  }
}

}  // namespace

extern "C" {

// Bytes of one rank's mailbox: 2 generations x world slots x (payload + flag line).
size_t simclr_comm_mailbox_bytes(int world, int max_doubles) {
  return (size_t)2 * world * ((size_t)max_doubles * 8 + kFlagBytes);
}

// Allocates and zeroes this rank's mailbox, returns its device address and the 64-byte hipIpc handle peers open.
// The library owns the allocation (simclr_comm_destroy).  Returns 3 when the runtime cannot export the allocation.
int simclr_comm_create(int world, int max_doubles, void** mailbox, void* ipc_handle_64) {
  SIMCLR_CHECK_ARG(world >= 1 && world <= 16 && max_doubles >= 1 && mailbox && ipc_handle_64,
                   "comm_create: world=%d (1..16), max_doubles=%d", world, max_doubles);
  const size_t bytes = simclr_comm_mailbox_bytes(world, max_doubles);
  void* ptr = nullptr;
  // uncached: flags and payload are written by OTHER devices / processes while kernels of this process poll them
  if (hipExtMallocWithFlags(&ptr, bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    if (hipMalloc(&ptr, bytes) != hipSuccess) {
      simclr_set_error("comm_create: cannot allocate %zu bytes", bytes);
      return 2;
    }
  }
  if (hipMemset(ptr, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(ptr);
    simclr_set_error("comm_create: memset failed");
    return 2;
  }
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, ptr) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(ptr);
    simclr_set_error("comm_create: hipIpcGetMemHandle failed (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)");
    return 3;
  }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpc handle size");
  memcpy(ipc_handle_64, &h, 64);
  *mailbox = ptr;
  return 0;
}

// Maps a peer's mailbox (its 64-byte handle) into this process.
int simclr_comm_open(const void* ipc_handle_64, void** mapped) {
  SIMCLR_CHECK_ARG(ipc_handle_64 && mapped, "comm_open: null argument");
  hipIpcMemHandle_t h;
  memcpy(&h, ipc_handle_64, 64);
  void* ptr = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    simclr_set_error("comm_open: hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
    return 3;
  }
  *mapped = ptr;
  return 0;
}
// Bound of the arrival wait of the exchanges launched AFTER this call, in seconds; <= 0: back to the default (600 s, or
// SIMCLR_PEER_STATS_TIMEOUT_S).  The set-up self-test of simclr_amd/comm.py runs with a short bound so that a mailbox that was
// mapped but does not deliver (another host's handle, a stale mapping) is found in seconds, not after the training-time bound.
static double g_timeout_s = 0.0;
int simclr_comm_set_timeout(double seconds) { g_timeout_s = seconds; return 0; }
int simclr_comm_close(void* mapped) { return hipIpcCloseMemHandle(mapped) == hipSuccess ? 0 : 3; }
int simclr_comm_destroy(void* mailbox) { return hipFree(mailbox) == hipSuccess ? 0 : 2; }

// out[i] = sum over ranks (in rank order) of in[i], i < count <= max_doubles.  peers: HOST array of `world` device
// addresses (peers[rank] = this rank's own mailbox, the others as returned by simclr_comm_open); seq: 1, 2, 3, ... the
// same on every rank for the same exchange; status (nullable, device int, zeroed once by the caller): STICKY count of peer
// arrivals that timed out so far -- never cleared here; an exchange that timed out returns NaN in `out`.
// in / out may alias.  One launch, one workgroup; enqueued on `stream`.  `seq` is a kernel argument: a captured hipGraph would
// replay a stale sequence number, so the call refuses a capturing stream (error 1).
int simclr_comm_stats_allreduce(const double* in, double* out, int count, void* const* peers, int rank, int world,
                                int max_doubles, unsigned seq, int* status, hipStream_t stream) {
  SIMCLR_CHECK_ARG(world >= 1 && world <= 16 && rank >= 0 && rank < world, "comm_stats_allreduce: rank %d / world %d", rank, world);
  SIMCLR_CHECK_ARG(count >= 1 && count <= max_doubles, "comm_stats_allreduce: count %d exceeds the slot capacity %d", count, max_doubles);
  SIMCLR_CHECK_ARG(seq != 0, "comm_stats_allreduce: sequence numbers start at 1");
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (stream && hipStreamIsCapturing(stream, &cap) == hipSuccess)
    SIMCLR_CHECK_ARG(cap == hipStreamCaptureStatusNone, "comm_stats_allreduce: the sequence number is a kernel argument; not capturable in a hipGraph");
  CommP p = {};
  for (int r = 0; r < world; ++r) {
    SIMCLR_CHECK_ARG(peers[r] != nullptr, "comm_stats_allreduce: peer %d not mapped", r);
    p.peer[r] = (unsigned char*)peers[r];
  }
  p.rank = rank; p.world = world; p.max_doubles = max_doubles; p.seq = seq;
  const double timeout_s = g_timeout_s > 0.0 ? g_timeout_s : (getenv("SIMCLR_PEER_STATS_TIMEOUT_S") ? atof(getenv("SIMCLR_PEER_STATS_TIMEOUT_S")) : 600.0);
  static const long long tick_hz = [] {            // wall_clock64() rate: hipDeviceAttributeWallClockRate is in kHz
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
    return (long long)khz * 1000;
  }();
  p.timeout_ticks = (long long)((timeout_s > 0.001 ? timeout_s : 0.001) * (double)tick_hz);
  hipLaunchKernelGGL(stats_exchange, dim3(1), dim3(256), 0, stream, p, in, out, count, status);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
