// orl_comm.hip - one-shot small-message SUM all-reduce over hipIpc-mapped peer memory (xGMI P2P) for gfx950.
//
// the PPO path exchanges ONE flat fp32 vector per optimiser step (both towers' raw
// gradient sums + denominators + logging sums, 38.9 KB at configuration 2).  At that size a ring collective is pure
// latency (2(G-1) dependent hops); here every rank PUSHES its vector once to every peer (G-1 independent P2P streams,
// one hop) and then sums the G vectors it received in RANK ORDER, so every rank computes the bit-identical result
// (replicas stay bit-identical without a broadcast) and the result does not depend on arrival order.
//
// Generalises the reference's only live collective, the sum-all-reduce + divide of
// openrl/modules/networks/utils/distributed_utils.py:22-26 (called on value-norm statistics) and the
// unimplemented DDP hook of openrl/algorithms/ppo.py:437-443.
//
// Transport (NCCL's "LL" idea, guide's data-tagged granules): the unit is one naturally aligned 8-byte granule
// {fp32 payload, 32-bit sequence tag} written with ONE system-scope store and read with system-scope loads; a
// reader polls a granule until its tag equals the collective's sequence number - no separate flag, no fence, no
// ordering assumption between granules.  Inboxes are double-buffered by sequence parity: a rank can only be one
// collective ahead of a peer (it needs the peer's contribution to finish), so parity k's slots are never rewritten
// before their reader is done.
//
// The same push / poll halves are fused into the PPO optimiser step (orl_ppo_reduce_pair_comm pushes the column sums
// it has just produced, orl_ppo_apply_comm polls while staging the raw sums), so a multi-GPU optimiser step is the same
// two launches as the single-GPU one (orl_apply.hip).
#include <string.h>
#include <new>
#include "orl_common.h"
#include "orl_comm.h"

namespace orl {

__global__ __launch_bounds__(256) void allreduce_small_kernel(CommDev C, float* __restrict__ data, int n) {
  const int stride = gridDim.x * blockDim.x;
  // push: my vector into slot [parity][my rank] of every peer's inbox
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = data[i];
    for (int p = 0; p < C.world; ++p)
      if (p != C.rank) comm_push(C, p, i, v);
  }
  // sum in rank order (own contribution straight from `data`)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) data[i] = comm_sum(C, i, data[i]);
}

}  // namespace orl

using namespace orl;

struct orl_comm {
  CommDev dev;                       // what the kernels get (by value)
  void* local;                       // my inbox (hipMalloc'ed here, exported over IPC)
  void* peer[ORL_COMM_MAX_WORLD];    // IPC-mapped inboxes (NULL for my own rank)
  int* err_dev;                      // device error word (poll timeout)
  size_t bytes;
  int connected;
  int device;
};

extern "C" {

int orl_comm_create(int rank, int world, int64_t capacity_floats, orl_comm** comm_out, unsigned char* handle_out) {
  ORL_REQUIRE(comm_out && handle_out, "orl_comm_create: null output");
  ORL_REQUIRE(world >= 1 && world <= ORL_COMM_MAX_WORLD && rank >= 0 && rank < world,
              "orl_comm_create: rank %d / world %d (max %d ranks: one node of MI355X)", rank, world, ORL_COMM_MAX_WORLD);
  ORL_REQUIRE(capacity_floats > 0 && capacity_floats <= (1 << 22), "orl_comm_create: capacity %lld floats",
              (long long)capacity_floats);
  ORL_REQUIRE(sizeof(hipIpcMemHandle_t) == ORL_IPC_HANDLE_BYTES, "orl_comm_create: hipIpcMemHandle_t is %zu bytes",
              sizeof(hipIpcMemHandle_t));
  orl_comm* c = new (std::nothrow) orl_comm();
  if (!c) return fail(ORL_E_INVALID, "orl_comm_create: out of host memory");
  memset(c, 0, sizeof(*c));
  const int64_t cap = (capacity_floats + 63) & ~(int64_t)63;
  c->bytes = (size_t)2 * world * cap * sizeof(unsigned long long);
  hipError_t e = hipGetDevice(&c->device);
  // fine-grained device memory: peers write it and the owner polls it WHILE kernels run on both sides, which is what
  // fine-grained coherence is specified for (coarse-grained memory is only guaranteed coherent at kernel boundaries)
  if (e == hipSuccess) {
    e = hipExtMallocWithFlags(&c->local, c->bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      e = hipMalloc(&c->local, c->bytes);
    }
  }
  if (e == hipSuccess) e = hipMemset(c->local, 0, c->bytes);  // tag 0 is never a live sequence number
  if (e == hipSuccess) e = hipMalloc((void**)&c->err_dev, sizeof(int));
  if (e == hipSuccess) e = hipMemset(c->err_dev, 0, sizeof(int));
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, c->local);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    if (c->local) (void)hipFree(c->local);
    if (c->err_dev) (void)hipFree(c->err_dev);
    delete c;
    return fail((int)e, "orl_comm_create: %s", hipGetErrorString(e));
  }
  memcpy(handle_out, &h, ORL_IPC_HANDLE_BYTES);
  c->dev.rank = rank; c->dev.world = world; c->dev.cap = (int)cap; c->dev.seq = 0; c->dev.err = c->err_dev;
  c->dev.inbox[rank] = (unsigned long long*)c->local;
  c->connected = world == 1;
  *comm_out = c;
  return 0;
}

int orl_comm_connect(orl_comm* c, const unsigned char* all_handles) {
  ORL_REQUIRE(c && all_handles, "orl_comm_connect: null pointer");
  for (int p = 0; p < c->dev.world; ++p) {
    if (p == c->dev.rank || c->peer[p]) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, all_handles + (size_t)p * ORL_IPC_HANDLE_BYTES, ORL_IPC_HANDLE_BYTES);
    void* ptr = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail((int)e, "orl_comm_connect: rank %d cannot map rank %d's inbox: %s", c->dev.rank, p,
                                     hipGetErrorString(e));
    c->peer[p] = ptr;
    c->dev.inbox[p] = (unsigned long long*)ptr;
  }
  c->connected = 1;
  return 0;
}

int orl_comm_destroy(orl_comm* c) {
  if (!c) return 0;
  (void)hipDeviceSynchronize();
  for (int p = 0; p < c->dev.world; ++p)
    if (c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
  if (c->local) (void)hipFree(c->local);
  if (c->err_dev) (void)hipFree(c->err_dev);
  delete c;
  return 0;
}

int orl_comm_error(orl_comm* c, void* stream) {
  ORL_REQUIRE(c, "orl_comm_error: null comm");
  int v = 0;
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e == hipSuccess) e = hipMemcpy(&v, c->err_dev, sizeof(int), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail((int)e, "orl_comm_error: %s", hipGetErrorString(e));
  if (v) return fail(ORL_E_INVALID, "orl_comm: rank %d timed out waiting for a peer's contribution", c->dev.rank);
  return 0;
}

int orl_comm_error_copy(orl_comm* c, int* err_out_dev, void* stream) {
  ORL_REQUIRE(c && err_out_dev, "orl_comm_error_copy: null pointer");
  const hipError_t e = hipMemcpyAsync(err_out_dev, c->err_dev, sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return fail((int)e, "orl_comm_error_copy: %s", hipGetErrorString(e));
  return 0;
}

int orl_allreduce_small(orl_comm* c, float* data, int n, void* stream) {
  ORL_REQUIRE(c && data, "orl_allreduce_small: null pointer");
  ORL_REQUIRE(c->connected, "orl_allreduce_small: orl_comm_connect has not run");
  ORL_REQUIRE(n > 0 && n <= c->dev.cap, "orl_allreduce_small: n=%d exceeds the comm's capacity %d", n, c->dev.cap);
  if (c->dev.world == 1) return 0;
  CommDev d;
  int rc = orl_comm_next(c, &d);
  if (rc) return rc;
  int grid = (n + 255) / 256;
  if (grid > 32) grid = 32;
  hipLaunchKernelGGL(allreduce_small_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d, data, n);
  return launch_status("orl_allreduce_small");
}

}  // extern "C"

// used by orl_apply.hip (same shared object): advance the sequence number and hand out the device view
int orl_comm_next(orl_comm* c, orl::CommDev* out) {
  if (!c || !out) return orl::fail(ORL_E_INVALID, "orl_comm_next: null");
  if (!c->connected) return orl::fail(ORL_E_INVALID, "orl_comm: orl_comm_connect has not run");
  c->dev.seq += 1;
  if (c->dev.seq == 0) c->dev.seq = 1;  // tag 0 = "never written"
  *out = c->dev;
  return 0;
}

// the device view of the collective orl_comm_next opened last (its second half: orl_ppo_apply_comm)
int orl_comm_current(orl_comm* c, orl::CommDev* out) {
  if (!c || !out) return orl::fail(ORL_E_INVALID, "orl_comm_current: null");
  if (c->dev.seq == 0) return orl::fail(ORL_E_INVALID, "orl_comm: no collective is open (push half has not run)");
  *out = c->dev;
  return 0;
}

int orl_comm_capacity(const orl_comm* c) { return c ? c->dev.cap : 0; }
