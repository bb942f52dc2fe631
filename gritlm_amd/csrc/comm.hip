// Cross-rank exchange of the pooled representations on RCCL, behind the C ABI (grit_comm_*).
//
// Replaces DistributedContrastiveLoss._dist_gather_tensor (gritlm/training/model.py:49-60): two list-API dist.all_gather calls, W + 1
// staging copies each and a torch.cat, issued on the compute stream after both towers are done.  Here one GROUPED collective gathers
// the query and the passage rows of every rank straight into the rank-major [W * n, H] matrices the loss kernel reads (rank r's rows
// land at [r * n, (r + 1) * n): the torch.cat order the targets arange(B) * G rely on, :45-46, :57-58) -- no packing copy, no cat --
// on a stream the CALLER supplies, e.g. a CU-masked side stream (grit_stream_create_cu_mask) so that the collective's copy kernels run
// beside the document tower's GEMMs instead of in front of the loss.  The backward of the gather is not a collective: each rank
// differentiates only its own rows (grit_infonce_rows_fwd_bwd row ranges).
//
// RCCL is resolved at run time (dlopen): under PyTorch the process already holds torch's librccl.so and the SAME instance is used
// (RTLD_NOLOAD first); a plain C host gets /opt/rocm's.  The library itself has no link-time dependency on RCCL: a build box without it
// still loads libgritlm_hip.so, and the grit_comm_* entry points then return GRIT_E_RCCL.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace grit {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

static const RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    for (const char* name : {"librccl.so", "librccl.so.1"}) {           // the instance the process already holds (PyTorch's), if any
      h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h)
      for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
      }
    if (!h) return;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
    api.GroupStart = (decltype(api.GroupStart))dlsym(h, "ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))dlsym(h, "ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd;
  });
  return api;
}

struct GritComm {
  ncclComm_t comm;
  int world, rank;
};

#define GRIT_RCCL(call, what)                                                                                       \
  do {                                                                                                              \
    const ncclResult_t r__ = (call);                                                                                \
    if (r__ != ncclSuccess) {                                                                                       \
      ::grit::set_error("%s: RCCL error %d (%s)", what, (int)r__, api.GetErrorString ? api.GetErrorString(r__) : "?"); \
      return GRIT_E_RCCL;                                                                                           \
    }                                                                                                               \
  } while (0)

}  // namespace grit

using namespace grit;

extern "C" {

int grit_comm_unique_id(void* id_out) {
  GRIT_REQUIRE(id_out, GRIT_E_BADARG, "grit_comm_unique_id: null pointer");
  const RcclApi& api = rccl();
  GRIT_REQUIRE(api.ok, GRIT_E_RCCL, "grit_comm_unique_id: librccl.so could not be loaded");
  ncclUniqueId id;
  GRIT_RCCL(api.GetUniqueId(&id), "grit_comm_unique_id");
  static_assert(sizeof(ncclUniqueId) == GRIT_COMM_ID_BYTES, "unique id size");
  memcpy(id_out, &id, sizeof(id));
  return GRIT_OK;
}

int grit_comm_init(const void* id, int world, int rank, void** comm_out) {
  GRIT_REQUIRE(id && comm_out, GRIT_E_BADARG, "grit_comm_init: null pointer");
  GRIT_REQUIRE(world > 0 && rank >= 0 && rank < world, GRIT_E_BADARG, "grit_comm_init: bad rank %d of %d", rank, world);
  const RcclApi& api = rccl();
  GRIT_REQUIRE(api.ok, GRIT_E_RCCL, "grit_comm_init: librccl.so could not be loaded");
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  GritComm* c = new GritComm{nullptr, world, rank};
  const ncclResult_t r = api.CommInitRank(&c->comm, world, uid, rank);       // on the calling thread's current HIP device
  if (r != ncclSuccess) {
    set_error("grit_comm_init: ncclCommInitRank failed: %d (%s)", (int)r, api.GetErrorString ? api.GetErrorString(r) : "?");
    delete c;
    return GRIT_E_RCCL;
  }
  *comm_out = c;
  return GRIT_OK;
}

int grit_comm_allgather_packed(void* comm, const float* q_local, int64_t nq_rows, const float* p_local, int64_t np_rows, int H, float* q_all,
                               float* p_all, void* stream) {
  GRIT_REQUIRE(comm, GRIT_E_BADARG, "grit_comm_allgather_packed: null communicator");
  GRIT_REQUIRE(H > 0 && nq_rows >= 0 && np_rows >= 0 && nq_rows + np_rows > 0, GRIT_E_BADARG, "grit_comm_allgather_packed: bad sizes");
  GRIT_REQUIRE((nq_rows == 0 || (q_local && q_all)) && (np_rows == 0 || (p_local && p_all)), GRIT_E_BADARG,
               "grit_comm_allgather_packed: null buffer");
  const RcclApi& api = rccl();
  GRIT_REQUIRE(api.ok, GRIT_E_RCCL, "grit_comm_allgather_packed: librccl.so could not be loaded");
  GritComm* c = (GritComm*)comm;
  hipStream_t st = (hipStream_t)stream;
  // both towers in ONE grouped operation: RCCL fuses the two gathers into a single launch per peer link
  GRIT_RCCL(api.GroupStart(), "grit_comm_allgather_packed");
  if (nq_rows) GRIT_RCCL(api.AllGather(q_local, q_all, (size_t)nq_rows * H, ncclFloat32, c->comm, st), "grit_comm_allgather_packed (q)");
  if (np_rows) GRIT_RCCL(api.AllGather(p_local, p_all, (size_t)np_rows * H, ncclFloat32, c->comm, st), "grit_comm_allgather_packed (p)");
  GRIT_RCCL(api.GroupEnd(), "grit_comm_allgather_packed");
  return GRIT_OK;
}

int grit_comm_destroy(void* comm) {
  if (!comm) return GRIT_OK;
  const RcclApi& api = rccl();
  GritComm* c = (GritComm*)comm;
  if (api.ok && c->comm) (void)api.CommDestroy(c->comm);
  delete c;
  return GRIT_OK;
}

// A stream whose kernels may only run on the first n_cus compute units of every XCD-interleaved mask word (hipExtStreamCreateWithCUMask):
// the side stream the collective is issued on, so that its copy kernels do not spread over the CUs the GEMMs are using.
int grit_stream_create_cu_mask(int n_cus, void** stream_out) {
  GRIT_REQUIRE(stream_out && n_cus > 0, GRIT_E_BADARG, "grit_stream_create_cu_mask: bad arguments");
  int dev = 0, total = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev);
  GRIT_REQUIRE(total > 0 && n_cus <= total, GRIT_E_BADARG, "grit_stream_create_cu_mask: %d of %d CUs", n_cus, total);
  uint32_t mask[16] = {0};
  const int words = (total + 31) / 32;
  // CU i of the mask is bit i; consecutive CU ids alternate over the XCDs, so the first n_cus bits take n_cus / 8 CUs from every XCD
  for (int i = 0; i < n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask) != hipSuccess) {
    (void)hipGetLastError();
    set_error("grit_stream_create_cu_mask: hipExtStreamCreateWithCUMask failed");
    return GRIT_E_LAUNCH;
  }
  *stream_out = st;
  return GRIT_OK;
}

int grit_stream_destroy(void* stream) {
  if (stream) (void)hipStreamDestroy((hipStream_t)stream);
  return GRIT_OK;
}

}  // extern "C"
