// Context, vectors, error plumbing of libstorm_hip.so.
#include <cstdarg>
#include <cstring>
#include <random>

#include <algorithm>
#include "common.hpp"

#include <cstddef>

namespace storm {

static thread_local char g_error[1024] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof g_error, fmt, ap);
  va_end(ap);
}

}  // namespace storm

using namespace storm;

static void pool_release(storm_hip_ctx *c);

extern "C" {

int storm_hip_abi_version(void) { return STORM_HIP_ABI_VERSION; }

const char *storm_hip_last_error(void) { return storm::g_error; }

int storm_hip_ctx_create(int device_id, storm_hip_ctx **out) {
  STORM_REQUIRE(out != nullptr, "ctx_create: out is null");
  *out = nullptr;
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0)
    STORM_FAIL(STORM_HIP_E_NO_DEVICE, "no HIP device available (%s)",
               e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  STORM_REQUIRE(device_id >= 0 && device_id < n_dev, "ctx_create: device %d out of range [0,%d)",
                device_id, n_dev);
  HIP_TRY(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    STORM_FAIL(STORM_HIP_E_NO_DEVICE, "device %d is %s; this library carries gfx950 code only",
               device_id, prop.gcnArchName);
  auto *c = new storm_hip_ctx();
  c->device = device_id;
  c->num_cus = prop.multiProcessorCount;
  c->name = prop.name;
  c->total_mem = (int64_t)prop.totalGlobalMem;
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  // (the two events that order the compute and the comm stream of ONE device: a device-scope release -- the default, a
  //  system-scope fence with cache write-back and invalidation, showed as 7 - 9 us in front of the launch behind the record)
  HIP_TRY(hipEventCreateWithFlags(&c->ev_x_ready, hipEventDisableTiming | hipEventReleaseToDevice));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_halo_done, hipEventDisableTiming | hipEventReleaseToDevice));
  HIP_TRY(hipEventCreate(&c->ev_t0));
  HIP_TRY(hipEventCreate(&c->ev_t1));
  c->partials_capacity = (int64_t)kMaxReduceBlocks * kMaxMulti * 2;  // (2 MiB; the widest user: ten sums x 16 384 blocks of mgs_multi_kernel)
  HIP_TRY(hipMalloc(&c->d_partials, sizeof(double) * (size_t)c->partials_capacity));
  HIP_TRY(hipMalloc(&c->d_partials2, sizeof(double) * kMaxMulti * kStage2));
  HIP_TRY(hipMalloc(&c->d_scalars, sizeof(double) * kMaxMulti));
  HIP_TRY(hipHostMalloc((void **)&c->h_scalars, sizeof(double) * kMaxMulti, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void **)&c->h_result_words, sizeof(unsigned long long) * 16 * 8, hipHostMallocMapped));
  memset(c->h_result_words, 0, sizeof(unsigned long long) * 16 * 8);
  HIP_TRY(hipHostGetDevicePointer((void **)&c->d_result_words, c->h_result_words, 0));
  HIP_TRY(hipMalloc((void **)&c->d_lat_slots, 2 * 256 * 256 + 256));  // latency.hip: all-reduce slots (two per block) + the gave-up flag
  HIP_TRY(hipMemset(c->d_lat_slots, 0, 2 * 256 * 256 + 256));
  HIP_TRY(hipMalloc((void **)&c->d_ticket_sums, sizeof(double) * 8 * 2048));  // [k <= 8][kTicketMaxGroups] group sums
  HIP_TRY(hipMalloc((void **)&c->d_tickets, sizeof(int) * (1 + 2048) * 16));  // ticket_device.hpp: kTicketMaxGroups, kTicketStride
  HIP_TRY(hipMemset(c->d_tickets, 0, sizeof(int) * (1 + 2048) * 16));
  HIP_TRY(hipMalloc((void **)&c->d_state, sizeof(SolverState)));
  HIP_TRY(hipMemset(c->d_state, 0, sizeof(SolverState)));
  HIP_TRY(hipHostMalloc((void **)&c->h_state, sizeof(SolverState), hipHostMallocMapped));
  HIP_TRY(hipHostMalloc((void **)&c->h_done_ring, sizeof(unsigned long long) * kStateRing, hipHostMallocMapped));
  HIP_TRY(hipHostGetDevicePointer((void **)&c->d_done_ring, c->h_done_ring, 0));
  *out = c;
  return STORM_HIP_OK;
}

int storm_hip_ctx_destroy(storm_hip_ctx *c) {
  if (!c) return STORM_HIP_OK;
  (void)hipSetDevice(c->device);
  c->lazy_q.clear();  // (statements nobody asked the result of)
  if (c->lazy_spare) (void)storm_hip_vec_destroy(c->lazy_spare), c->lazy_spare = nullptr;
  (void)hipDeviceSynchronize();
  comm_destroy(c);
  for (auto &ev : c->ev_ring) (void)hipEventDestroy(ev);
  for (auto &r : c->krylov_free) {
    if (r.S) (void)hipFree(r.S);
    (void)hipFree(r.d_st);
    (void)hipHostFree(r.h_st);
    (void)hipHostFree(r.h_ring);
  }
  for (auto &ev : c->prof_events) (void)hipEventDestroy(ev);
  pool_release(c);
  for (auto &a : c->arenas) (void)hipFree(a.base);
  (void)hipFree(c->d_partials);
  if (c->d_gmres) (void)hipFree(c->d_gmres);
  (void)hipFree(c->d_partials2);
  (void)hipFree(c->d_scalars);
  (void)hipFree(c->d_lat_slots);
  if (c->d_res_exch) (void)hipFree(c->d_res_exch);
  if (c->d_res_slots) (void)hipFree(c->d_res_slots);
  if (c->d_quad_slots) (void)hipFree(c->d_quad_slots);
  if (c->d_res_prof) (void)hipFree(c->d_res_prof);
  (void)hipFree(c->d_tickets);
  (void)hipFree(c->d_ticket_sums);
  (void)hipHostFree(c->h_scalars);
  (void)hipHostFree(c->h_result_words);
  (void)hipFree(c->d_state);
  (void)hipHostFree(c->h_state);
  (void)hipHostFree(c->h_done_ring);
  (void)hipEventDestroy(c->ev_x_ready);
  (void)hipEventDestroy(c->ev_halo_done);
  (void)hipEventDestroy(c->ev_t0);
  (void)hipEventDestroy(c->ev_t1);
  (void)hipStreamDestroy(c->stream);
  (void)hipStreamDestroy(c->comm_stream);
  delete c;
  return STORM_HIP_OK;
}

int storm_hip_ctx_sync(storm_hip_ctx *c) {
  STORM_REQUIRE(c, "ctx_sync: null context");
  STORM_TRY(lazy_sync(c));
  HIP_TRY(hipStreamSynchronize(c->comm_stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return comm_check_error(c);
}

int storm_hip_ctx_info(storm_hip_ctx *c, char *name, int name_len, int *num_cus,
                       int64_t *total_mem_bytes) {
  STORM_REQUIRE(c, "ctx_info: null context");
  if (name && name_len > 0) {
    strncpy(name, c->name.c_str(), (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  if (num_cus) *num_cus = c->num_cus;
  if (total_mem_bytes) *total_mem_bytes = c->total_mem;
  return STORM_HIP_OK;
}

int storm_hip_ctx_set_option(storm_hip_ctx *c, const char *key, int64_t value) {
  STORM_REQUIRE(c && key, "ctx_set_option: null argument");
  if (!strcmp(key, "ell_cap")) c->opt_ell_cap = value;
  else if (!strcmp(key, "generic_solvers")) c->opt_generic_solvers = value;
  else if (!strcmp(key, "rccl_fused")) c->opt_rccl_fused = value;
  else if (!strcmp(key, "rccl_ticket")) c->opt_rccl_ticket = value;
  else if (!strcmp(key, "rccl_early_halo")) c->opt_rccl_early_halo = value;
  else if (!strcmp(key, "rccl_flag_wait")) c->opt_rccl_flag_wait = value;
  else if (!strcmp(key, "comm_wait_seconds")) {
    STORM_REQUIRE(value >= 1 && value <= 3600, "ctx_set_option: comm_wait_seconds %lld (1 .. 3600)", (long long)value);
    c->opt_comm_wait_seconds = value;
  }
  else if (!strcmp(key, "latency_path")) c->opt_latency_path = value;
  else if (!strcmp(key, "resident_path")) c->opt_resident_path = value;
  else if (!strcmp(key, "resident_max_rows")) c->opt_resident_max_rows = value;
  else if (!strcmp(key, "resident_planes")) c->opt_resident_planes = value;
  else if (!strcmp(key, "resident_profile")) c->opt_resident_profile = value;
  else if (!strcmp(key, "coop_force_fail")) c->opt_coop_force_fail = value;
  else if (!strcmp(key, "coop_mgs")) c->opt_coop_mgs = value;
  else if (!strcmp(key, "coop_mgs_lds")) c->opt_coop_mgs_lds = value;
  else if (!strcmp(key, "coop_mgs_quad")) c->opt_coop_mgs_quad = value;
  else if (!strcmp(key, "cg_march_fill")) c->opt_cg_march_fill = value;
  else if (!strcmp(key, "resident_early")) c->opt_resident_early = value;
  else if (!strcmp(key, "mgs_steps")) c->opt_mgs_steps = value;
  else if (!strcmp(key, "vec_arena")) c->opt_vec_arena = value;
  else if (!strcmp(key, "spmv_mixed")) c->opt_spmv_mixed = value;
  else if (!strcmp(key, "spmv_canon_tile")) c->opt_spmv_canon_tile = value;
  else if (!strcmp(key, "spmv_canon_tile_min_rows")) c->opt_spmv_canon_tile_min_rows = value;
  else if (!strcmp(key, "fused_reduce")) c->opt_fused_reduce = (int)value;
  else if (!strcmp(key, "ticket_reduce")) c->opt_ticket_reduce = (int)value;
  else if (!strcmp(key, "ticket_verify")) c->opt_ticket_verify = value;
  else if (!strcmp(key, "ticket_verify_inject")) c->opt_ticket_verify_inject = value;
  else if (!strcmp(key, "lin_fuse")) c->opt_lin_fuse = (int)value;
  else if (!strcmp(key, "latency_rows")) c->opt_latency_rows = value;
  else if (!strcmp(key, "latency_cache")) c->opt_latency_cache = (int)value;
  else if (!strcmp(key, "nontemporal")) c->opt_nt = value;
  else if (!strcmp(key, "spmv_dict")) c->opt_spmv_dict = value;
  else if (!strcmp(key, "pool_bytes")) {
    c->opt_pool_bytes = value;
    // trim now; 0 = give everything back at once -- idle arenas too, which pool_bytes does not count (a context
    // whose pool holds only arena slots has pool_bytes == 0)
    if ((int64_t)c->pool_bytes > value || (value == 0 && !c->pool.empty())) {
      (void)hipStreamSynchronize(c->stream);
      pool_release(c);
    }
  }
  else if (!strcmp(key, "spmv_spw")) c->opt_spmv_spw = value;
  else if (!strcmp(key, "spmv_xcd_remap")) {
    STORM_REQUIRE(value >= 0 && value <= 4096, "ctx_set_option: spmv_xcd_remap %lld (0 .. 4096)", (long long)value);
    c->opt_spmv_xcd_remap = c->opt_spmv_xcd_remap_sell = value;  // (A/B knob: every kernel's run length at once)
  }
  else if (!strcmp(key, "test_disable")) {
    // TEST HOOK, not an option: every bit switches one refinement of a kernel off, so that a test can run the refined and
    // the plain form side by side and demand the same bits (the plain forms exist for that purpose only).
    //   1  GMRES chain kernel applies the operator itself (0: the apply is a launch in front of the chain)
    //   2  the chain's prefetch of the next group's basis vectors under its all-reduce (registers / LDS-DMA)
    //   4  resident path: a row pair's coefficients kept in registers from plane to plane
    //   8  resident CG: half of the waves form the halo of the new direction before their own update
    //  16  latency path: rows published with awaited atomic exchanges (off: write-through stores)
    //  32  one-kernel paths launched like any kernel (off: hipLaunchCooperativeKernel)
    //  64  marching CG step: odd z-chunks march downwards
    // 128  GMRES chain kernel: one contiguous run of row chunks per XCD (off: chunk = block index)
    // 256  GMRES chain kernel: the order of the basis vectors alternates with k (off: ascending, the reference's)
    // 512  GMRES chain kernel: the column's earlier rotations under the norm's all-reduce (off: after everything else)
    c->opt_test_disable = value;
    c->opt_coop_mgs_apply = !(value & 1), c->opt_coop_mgs_prefetch = c->opt_coop_mgs_lds_prefetch = !(value & 2);
    c->opt_resident_apply_cache = !(value & 4), c->opt_resident_halo_interleave = !(value & 8);
    c->opt_latency_publish = !(value & 16), c->opt_coop_plain = !(value & 32), c->opt_cg_march_alternate = !(value & 64);
    c->opt_coop_mgs_xcd_runs = !(value & 128), c->opt_coop_mgs_alternate = !(value & 256);
    c->opt_coop_mgs_rotate_early = !(value & 512);
  }
  else if (!strcmp(key, "lazy_statements")) {
    if (value == 0) {
      STORM_TRY(lazy_sync(c));
      // (the spare vector of the fused CG step goes back to the pool: the next host loop finds it there)
      if (c->lazy_spare) (void)storm_hip_vec_destroy(c->lazy_spare), c->lazy_spare = nullptr;
    }
    c->opt_lazy = value;
  }
  else if (!strcmp(key, "profile_spmv")) c->opt_profile_spmv = value;
  else if (!strcmp(key, "profile_comm")) {
    c->opt_profile_comm = value;
    if (value != 0) STORM_TRY(comm_profile_reset(c));
  }
  else if (!strcmp(key, "cg_fuse")) c->opt_cg_fuse = value;
  else if (!strcmp(key, "cg_march")) c->opt_cg_march = value;
  else if (!strcmp(key, "blas1_nt")) c->opt_blas1_nt = value;
  else STORM_FAIL(STORM_HIP_E_INVALID, "ctx_set_option: unknown key '%s'", key);
  return STORM_HIP_OK;
}

int storm_hip_ctx_get_counter(storm_hip_ctx *c, const char *key, int64_t *value) {
  STORM_REQUIRE(c && key && value, "ctx_get_counter: null argument");
  if (!strcmp(key, "resident_solves")) *value = c->n_resident_solves;
  else if (!strcmp(key, "latency_solves")) *value = c->n_latency_solves;
  else if (!strcmp(key, "throughput_solves")) *value = c->n_throughput_solves;
  else if (!strcmp(key, "engine_solves")) *value = c->n_engine_solves;
  else if (!strcmp(key, "cg_fused_steps")) *value = c->n_cg_fused_steps;
  else if (!strcmp(key, "lazy_fused_dots")) *value = c->n_lazy_fused_dots;
  else if (!strcmp(key, "lazy_fused_pairs")) *value = c->n_lazy_fused_pairs;
  else if (!strcmp(key, "lazy_apply_dots")) *value = c->n_lazy_apply_dots;
  else if (!strcmp(key, "lazy_cg_steps")) *value = c->n_lazy_cg_steps;
  else if (!strcmp(key, "lazy_waiting")) *value = (int64_t)c->lazy_q.size();
  else if (!strncmp(key, "ipc_", 4)) {
    // the peer-window transport's device-side waits (csrc/ipc_device.hpp IpcDev::stat): ticks of 10 ns and counts
    static const char *names[6] = {"ipc_allreduce_wait_ticks", "ipc_allreduces", "ipc_ack_wait_ticks", "ipc_ack_waits",
                                   "ipc_halo_slow_poll_ticks", "ipc_halo_slow_polls"};
    int k = -1;
    for (int i = 0; i < 6; ++i)
      if (!strcmp(key, names[i])) k = i;
    STORM_REQUIRE(k >= 0, "ctx_get_counter: unknown key '%s'", key);
    const long long v = comm_ipc_stat(c, k);
    STORM_REQUIRE(v >= 0, "ctx_get_counter: '%s' needs the peer-window transport", key);
    *value = v;
  }
  else if (!strncmp(key, "rccl_prof_", 10)) {
    // option profile_comm on the RCCL transport (csrc/comm.hip): sums over the stamped exchanges / all-reduces, ticks of 10 ns
    static const char *names[8] = {"rccl_prof_exchanges", "rccl_prof_event_to_comm_ticks", "rccl_prof_pack_ticks",
                                   "rccl_prof_sendrecv_ticks", "rccl_prof_unhidden_wait_ticks", "rccl_prof_resume_ticks",
                                   "rccl_prof_allreduces", "rccl_prof_allreduce_ticks"};
    int k = -1;
    for (int i = 0; i < 8; ++i)
      if (!strcmp(key, names[i])) k = i;
    STORM_REQUIRE(k >= 0, "ctx_get_counter: unknown key '%s'", key);
    const long long v = comm_profile_read(c, k);
    STORM_REQUIRE(v >= 0, "ctx_get_counter: '%s' needs the RCCL transport with option profile_comm = 1", key);
    *value = v;
  }
  else if (!strncmp(key, "resident_phase_max_", 19) || !strncmp(key, "resident_phase_mean_", 20)) {
    // (option resident_profile: ticks of 10 ns that the last resident solve's blocks spent in phase k of their loop)
    const bool mx = key[15] == 'm' && key[16] == 'a';
    const int k = atoi(key + (mx ? 19 : 20));
    STORM_REQUIRE(c->d_res_prof != nullptr && c->res_prof_blocks > 0 && k >= 0 && k < 8, "ctx_get_counter: no resident profile (option resident_profile)");
    std::vector<long long> h((size_t)c->res_prof_blocks * 8);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(h.data(), c->d_res_prof, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    long long m = 0, sum = 0;
    for (int b = 0; b < c->res_prof_blocks; ++b) m = std::max(m, h[(size_t)b * 8 + k]), sum += h[(size_t)b * 8 + k];
    *value = mx ? m : sum / c->res_prof_blocks;
  } else if (!strncmp(key, "resident_phase_block_", 21)) {  // "resident_phase_block_<b>_<k>": block b's ticks in phase k
    int b = 0, k = 0;
    STORM_REQUIRE(sscanf(key + 21, "%d_%d", &b, &k) == 2 && c->d_res_prof != nullptr && b >= 0 && b < c->res_prof_blocks && k >= 0 && k < 8,
                  "ctx_get_counter: no such block / phase");
    long long v = 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(&v, c->d_res_prof + (size_t)b * 8 + k, sizeof v, hipMemcpyDeviceToHost));
    *value = v;
  }
  else STORM_FAIL(STORM_HIP_E_INVALID, "ctx_get_counter: unknown key '%s'", key);
  return STORM_HIP_OK;
}

int storm_hip_ctx_get_spmv_profile(storm_hip_ctx *c, int64_t *launches, double *total_ms, double *min_ms) {
  STORM_REQUIRE(c, "get_spmv_profile: null context");
  HIP_TRY(hipStreamSynchronize(c->stream));
  double total = 0.0, mn = 0.0;
  const size_t pairs = c->prof_used / 2;
  for (size_t i = 0; i < pairs; ++i) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->prof_events[2 * i], c->prof_events[2 * i + 1]));
    total += ms;
    if (i == 0 || ms < mn) mn = ms;
  }
  if (launches) *launches = (int64_t)pairs;
  if (total_ms) *total_ms = total;
  if (min_ms) *min_ms = mn;
  c->prof_used = 0;
  return STORM_HIP_OK;
}

int storm_hip_ctx_get_spmv_profile_samples(storm_hip_ctx *c, double *ms_out, int64_t capacity, int64_t *count) {
  STORM_REQUIRE(c && count && (ms_out || capacity == 0), "get_spmv_profile_samples: null argument");
  HIP_TRY(hipStreamSynchronize(c->stream));
  const size_t pairs = c->prof_used / 2;
  for (size_t i = 0; i < pairs && (int64_t)i < capacity; ++i) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->prof_events[2 * i], c->prof_events[2 * i + 1]));
    ms_out[i] = ms;
  }
  *count = (int64_t)pairs;
  c->prof_used = 0;
  return STORM_HIP_OK;
}

int storm_hip_timer_start(storm_hip_ctx *c) {
  STORM_REQUIRE(c, "timer_start: null context");
  STORM_TRY(lazy_sync(c));
  HIP_TRY(hipEventRecord(c->ev_t0, c->stream));
  return STORM_HIP_OK;
}

int storm_hip_timer_stop(storm_hip_ctx *c, float *elapsed_ms) {
  STORM_REQUIRE(c && elapsed_ms, "timer_stop: null argument");
  STORM_TRY(lazy_sync(c));
  HIP_TRY(hipEventRecord(c->ev_t1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev_t1));
  HIP_TRY(hipEventElapsedTime(elapsed_ms, c->ev_t0, c->ev_t1));
  return STORM_HIP_OK;
}

// ---- vectors ---------------------------------------------------------------------

}  // extern "C"
enum { kZeroAll = 0, kZeroEdges = 1, kZeroNothing = 2 };
static int vec_create_impl(storm_hip_ctx *c, int64_t n_owned, int64_t n_halo, storm_hip_vec **out, int zero_mode);
extern "C" int storm_hip_vec_create(storm_hip_ctx *c, int64_t n_owned, int64_t n_halo, storm_hip_vec **out) {
  return vec_create_impl(c, n_owned, n_halo, out, kZeroAll);
}
namespace storm {
int ring_post(storm_hip_ctx *c, std::vector<hipEvent_t> &events, int64_t it) {
  if (c->opt_poll_events == 0) return STORM_HIP_OK;
  while (events.size() < (size_t)kStateRing) {
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    events.push_back(ev);
  }
  HIP_TRY(hipEventRecord(events[(size_t)(it % kStateRing)], c->stream));
  return STORM_HIP_OK;
}

// The verdict of iteration index `it` (0-based; the device counts from 1).
int ring_wait(storm_hip_ctx *c, std::vector<hipEvent_t> &events, volatile unsigned long long *ring, int64_t it, bool *stop) {
  volatile unsigned long long *w = ring + it % kStateRing;
  const unsigned long long want = (unsigned long long)(it + 1);
  const unsigned long long gen = c->ring_gen & 0xfffffull;
  auto posted = [&](bool *s) {  // (a word of another generation -- a late post of an aborted solve -- is not this solve's)
    const unsigned long long v = *w;
    if ((v >> 44) != gen) return false;
    const unsigned long long i = (v >> 1) & kRingIterMask;
    if (i == kRingIterMask) return *s = true, true;  // begin(): nothing to iterate
    if (i == want) return *s = (v & 1ull) != 0, true;
    return false;
  };
  *stop = false;
  if (c->opt_poll_events != 0 && events.size() == (size_t)kStateRing) {  // (the option switched on mid-solve: poll)
    HIP_TRY(hipEventSynchronize(events[(size_t)(it % kStateRing)]));
    (void)posted(stop);
    return STORM_HIP_OK;
  }
  for (unsigned long spin = 1;; ++spin) {
    if (posted(stop)) return STORM_HIP_OK;
    if ((spin & 0x3ffful) == 0) {
      const hipError_t e = hipStreamQuery(c->stream);
      if (e == hipSuccess) {  // everything enqueued has run: an iteration that posted nothing has not ended the solve
        (void)posted(stop);
        return STORM_HIP_OK;
      }
      if (e != hipErrorNotReady) HIP_TRY(e);
    } else {
      __builtin_ia32_pause();
    }
  }
}

// A solver's WORK vector whose owned rows the solver writes before it reads them (CG's r, p, z; BiCGStab's r, rt, p, v,
// t): only the guard in front, the halo tail and the padding behind are zeroed -- three 134 MB memsets per 256^3 CG
// solve less (~90 us; a K = 20 solve is 4.7 ms).
int vec_create_work(const storm_hip_vec *like, storm_hip_vec **out) {
  STORM_REQUIRE(like, "vec_create_work: null vector");
  return vec_create_impl(like->ctx, like->n_owned, like->n_halo, out, kZeroEdges);
}

constexpr int kEdgeBatch = 8;
struct EdgePtrs {
  double *base[kEdgeBatch];
};
// the guard in front of element 0 and everything behind the owned rows (halo rows, padding) of `gridDim.y` vectors
__global__ __launch_bounds__(kBlock) void zero_edges_kernel(EdgePtrs v, int64_t tail0, int64_t total) {
  double *b = v.base[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < kVecGuard + (total - tail0); i += (int64_t)gridDim.x * kBlock)
    b[i < kVecGuard ? i : tail0 + (i - kVecGuard)] = 0.0;
}
int vec_create_work_batch(const storm_hip_vec *like, int count, storm_hip_vec **out) {
  STORM_REQUIRE(like && out && count >= 0, "vec_create_work_batch: bad argument");
  storm_hip_ctx *c = like->ctx;
  for (int i0 = 0; i0 < count; i0 += kEdgeBatch) {
    const int k = std::min(kEdgeBatch, count - i0);
    EdgePtrs e{};
    for (int i = 0; i < k; ++i) {
      const int st = vec_create_impl(c, like->n_owned, like->n_halo, &out[i0 + i], kZeroNothing);
      if (st != STORM_HIP_OK) {
        for (int j = 0; j < i0 + i; ++j) (void)storm_hip_vec_destroy(out[j]), out[j] = nullptr;
        return st;
      }
      e.base[i] = out[i0 + i]->base;
    }
    const int64_t total = (int64_t)(out[i0]->bytes / sizeof(double)), tail0 = kVecGuard + like->n_owned;
    const int64_t work = kVecGuard + (total - tail0);
    hipLaunchKernelGGL(zero_edges_kernel, dim3((unsigned)std::min<int64_t>((work + kBlock - 1) / kBlock, 1024), (unsigned)k), dim3(kBlock), 0,
                       c->stream, e, tail0, total);
    HIP_TRY(hipGetLastError());
  }
  return STORM_HIP_OK;
}

__global__ void state_init_kernel(SolverState *st, double abs_tol, double rel_tol, long long num_iterations, double *history,
                                  unsigned long long *ring, unsigned long long gen) {
  for (int i = threadIdx.x; i < kSlab; i += blockDim.x) st->s[i] = 0.0;
  if (threadIdx.x != 0) return;
  st->initial_error = st->absolute_error = st->relative_error = 0.0;
  st->abs_tol = abs_tol, st->rel_tol = rel_tol;
  st->iteration = 0, st->num_iterations = num_iterations;
  st->done = 0, st->converged = 0, st->verify_failed = 0;
  st->history = history, st->done_ring = ring, st->ring_gen = gen;
}
// (the named fields behind the scalar slab: the host never reads the slab, and 2 KB of 8-byte stores over PCIe cost
//  ~16 us -- the stepping interface reads the state once per iteration)
__global__ void state_export_kernel(const SolverState *st, SolverState *host) {
  constexpr size_t first = offsetof(SolverState, initial_error) / sizeof(unsigned long long);
  static_assert(offsetof(SolverState, initial_error) % sizeof(unsigned long long) == 0 &&
                    sizeof(SolverState) % sizeof(unsigned long long) == 0 &&
                    sizeof(SolverState) / sizeof(unsigned long long) - first <= 64,
                "the fields behind SolverState::s are copied as 8-byte words by one wavefront");
  const unsigned long long *s = reinterpret_cast<const unsigned long long *>(st);
  unsigned long long *d = reinterpret_cast<unsigned long long *>(host);
  const size_t i = first + threadIdx.x;
  if (i < sizeof(SolverState) / sizeof(unsigned long long)) d[i] = s[i];
}
int state_read(storm_hip_ctx *c, const SolverState *d_state, SolverState *h_pinned) {
  SolverState *h_dev = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void **)&h_dev, h_pinned, 0));
  hipLaunchKernelGGL(state_export_kernel, dim3(1), dim3(kWave), 0, c->stream, d_state, h_dev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  return STORM_HIP_OK;
}
int state_init(storm_hip_ctx *c, SolverState *d_state, double abs_tol, double rel_tol, long long num_iterations, double *history,
               unsigned long long *d_ring) {
  c->ring_gen = (c->ring_gen + 1) & 0xfffffull;
  if (c->ring_gen == 0) c->ring_gen = 1;  // (0: the ring's cleared words)
  hipLaunchKernelGGL(state_init_kernel, dim3(1), dim3(kBlock), 0, c->stream, d_state, abs_tol, rel_tol, num_iterations, history, d_ring,
                     c->ring_gen);
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}
}  // namespace storm
static bool in_arena(const storm_hip_ctx *c, const void *p) {
  for (const auto &a : c->arenas)
    if ((const char *)p >= a.base && (const char *)p < a.base + (size_t)a.slots * a.pitch) return true;
  return false;
}
// Give pooled storage back to the driver (an allocation failed, or the pool budget shrank; the caller has synchronised
// the stream): every vector allocated by itself, and every ARENA whose slots are all back in the pool -- an arena with a
// slot still in use stays, with its free slots pooled.  pool_bytes counts the stand-alone vectors only: arena slots have
// the arenas' own budget (vec_arena_max_bytes) and must not crowd the others out of the pool.
static void pool_release(storm_hip_ctx *c) {
  std::vector<int> pooled(c->arenas.size(), 0);
  auto arena_of = [&](const void *p) {
    for (size_t i = 0; i < c->arenas.size(); ++i) {
      const auto &a = c->arenas[i];
      if ((const char *)p >= a.base && (const char *)p < a.base + (size_t)a.slots * a.pitch) return (int)i;
    }
    return -1;
  };
  for (auto &pb : c->pool) {
    const int i = arena_of(pb.second);
    if (i >= 0) ++pooled[(size_t)i];
  }
  std::vector<std::pair<size_t, double *>> keep;
  for (auto &pb : c->pool) {
    const int i = arena_of(pb.second);
    if (i < 0) (void)hipFree(pb.second);
    else if (pooled[(size_t)i] < c->arenas[(size_t)i].used) keep.push_back(pb);  // (its arena still has a slot in use)
  }
  std::vector<storm_hip_ctx::VecArena> live;
  for (size_t i = 0; i < c->arenas.size(); ++i) {
    if (c->arenas[i].used > 0 && pooled[i] == c->arenas[i].used) (void)hipFree(c->arenas[i].base);  // idle: every handed-out slot is back
    else live.push_back(c->arenas[i]);
  }
  c->arenas.swap(live);
  c->pool.swap(keep), c->pool_bytes = 0;
}
// A slot of an arena of this size class (a new arena when all are taken); null: no arena (too small, too large, off,
// or the allocation failed) -- the caller allocates the vector by itself.
static double *arena_take(storm_hip_ctx *c, size_t bytes) {
  if (c->opt_vec_arena == 0 || bytes < ((size_t)1 << 20)) return nullptr;
  // The distance between two vectors: the smallest one = 2 MiB (mod 4 MiB) that holds the vector.  Measured on the
  // 256^3 CG (tools/arena_sweep.py, vectors of 128 MiB + 288 B): 130, 131, 133, 134, 138, 142 MiB apart 4 500 - 4 550
  // it/s; 132 and 136 MiB apart -- multiples of 4 MiB; 132 MiB is where separate hipMalloc calls put them -- 4 360 - 4 380.
  constexpr size_t kMiB = (size_t)1 << 20;
  size_t pitch = (bytes + 4 * kMiB - 1) / (4 * kMiB) * (4 * kMiB);  // a multiple of 4 MiB
  pitch = pitch - 2 * kMiB >= bytes ? pitch - 2 * kMiB : pitch + 2 * kMiB;
  pitch += (size_t)c->opt_vec_arena_skew_kib * 1024;
  int slots = (int)std::max<int64_t>(2, c->opt_vec_arena_slots);
  for (auto &a : c->arenas) {
    if (a.bytes != bytes || a.pitch != pitch) continue;
    if (a.used < a.slots) return reinterpret_cast<double *>(a.base + (size_t)(a.used++) * pitch);
    slots = std::min(64, std::max(slots, 2 * a.slots));  // (a solver with a long basis: the next arena twice as long)
  }
  while (slots > 1 && (size_t)slots * pitch > ((size_t)12 << 30)) slots /= 2;  // (an arena of at most 12 GiB)
  if (slots < 2) return nullptr;
  size_t held = 0;
  for (const auto &h : c->arenas) held += (size_t)h.slots * h.pitch;
  if (held + (size_t)slots * pitch > (size_t)c->opt_vec_arena_max_bytes) return nullptr;  // (many sizes in one context: enough reserved)
  storm_hip_ctx::VecArena a;
  a.bytes = bytes, a.pitch = pitch, a.slots = slots, a.used = 1;
  const size_t total = (size_t)slots * pitch;
  if (c->opt_vec_arena_contiguous == 0 || hipExtMallocWithFlags((void **)&a.base, total, hipDeviceMallocContiguous) != hipSuccess) {
    (void)hipGetLastError();
    if (hipMalloc((void **)&a.base, total) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
  }
  c->arenas.push_back(a);
  return reinterpret_cast<double *>(a.base);
}


static int vec_create_impl(storm_hip_ctx *c, int64_t n_owned, int64_t n_halo, storm_hip_vec **out, int zero_mode) {
  STORM_REQUIRE(c && out, "vec_create: null argument");
  *out = nullptr;
  STORM_REQUIRE(n_owned >= 0 && n_halo >= 0, "vec_create: negative size");
  STORM_REQUIRE(n_owned + n_halo < (int64_t)INT32_MAX, "vec_create: %lld rows exceed int32 indexing",
                (long long)(n_owned + n_halo));
  auto *v = new storm_hip_vec();
  v->ctx = c;
  v->n_owned = n_owned;
  v->n_halo = n_halo;
  // Round the allocation up so 16-byte vector accesses of the last rows stay in bounds, and keep a
  // zero-filled guard of kVecGuard doubles IN FRONT of element 0 (256 bytes: the alignment of `d` is
  // unchanged): the paired-row SpMV reads x[col], x[col + 1] with one 16-byte load per slot, and where one
  // row of a pair has no neighbour in a slot (weight 0) that load may touch x[-1] or x[n].  The guard and
  // the tail padding are zero and never written, so such a term is exactly 0 * (0 - x_i).
  const size_t bytes = sizeof(double) * (size_t)(kVecGuard + (n_owned + n_halo + 3) / 4 * 4 + 4);
  v->bytes = bytes;
  hipError_t e = hipSuccess;
  double *base = nullptr;
  // last in, first out, order kept: a solver that releases its work vectors in the reverse order of their creation
  // (VecPool, solvers.hip) gets every vector back in the same role next time -- until round 4 the roles permuted with
  // period two, and the 256^3 CG rate alternated between 4 588 and 4 660 it/s with them (tools/rate_stability.py)
  for (size_t i = c->pool.size(); i-- > 0;) {
    if (c->pool[i].first == bytes) {
      base = c->pool[i].second;
      if (!in_arena(c, base)) c->pool_bytes -= bytes;
      c->pool.erase(c->pool.begin() + (std::ptrdiff_t)i);
      break;
    }
  }
  // (option vec_contiguous: physically contiguous storage -- the largest page-table fragments the driver can give)
  auto device_malloc = [&](double **p) {
    if (c->opt_vec_contiguous != 0 && bytes >= ((size_t)1 << 21)) {
      if (hipExtMallocWithFlags((void **)p, bytes, hipDeviceMallocContiguous) == hipSuccess) return hipSuccess;
      (void)hipGetLastError();
    }
    return hipMalloc(p, bytes);
  };
  if (base == nullptr) base = arena_take(c, bytes);
  if (base == nullptr) {
    e = device_malloc(&base);
    if (e != hipSuccess && !c->pool.empty()) {  // give the pooled storage back and retry
      (void)hipStreamSynchronize(c->stream);
      pool_release(c);
      e = device_malloc(&base);
    }
  }
  if (e != hipSuccess) {
    delete v;
    STORM_FAIL(STORM_HIP_E_ALLOC, "vec_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  v->base = base;
  v->d = base + kVecGuard;
  // Field::assign value-initialises (Feathers/Field.hpp:82-84): zero fill.  (Work vectors: everything but the owned rows.)
  if (zero_mode == kZeroAll) {
    e = hipMemsetAsync(v->base, 0, bytes, c->stream);
  } else if (zero_mode == kZeroEdges) {
    e = hipMemsetAsync(v->base, 0, sizeof(double) * kVecGuard, c->stream);
    const size_t tail0 = sizeof(double) * (size_t)(kVecGuard + n_owned);
    if (e == hipSuccess && bytes > tail0) e = hipMemsetAsync(reinterpret_cast<char *>(v->base) + tail0, 0, bytes - tail0, c->stream);
  }
  if (e != hipSuccess) {
    if (!in_arena(c, v->base)) (void)hipFree(v->base);
    else c->pool.emplace_back(bytes, v->base);  // (an arena slot goes back to the pool, not into the void)
    delete v;
    STORM_FAIL(STORM_HIP_E_HIP, "vec_create: memset failed: %s", hipGetErrorString(e));
  }
  *out = v;
  return STORM_HIP_OK;
}

extern "C" {

int storm_hip_vec_create_like(const storm_hip_vec *other, storm_hip_vec **out) {
  STORM_REQUIRE(other, "vec_create_like: null vector");
  return storm_hip_vec_create(other->ctx, other->n_owned, other->n_halo, out);
}

int storm_hip_vec_destroy(storm_hip_vec *v) {
  if (!v) return STORM_HIP_OK;
  storm_hip_ctx *c = v->ctx;
  (void)lazy_sync(c);  // (a waiting statement may read or write this storage)
  const bool arena_slot = v->base && in_arena(c, v->base);
  if (v->base && (arena_slot || (int64_t)(c->pool_bytes + v->bytes) <= c->opt_pool_bytes)) {
    // later users of this storage are ordered behind its pending kernels by the compute stream; the
    // comm stream only touches a vector between two events of one SpMV (comm.hip)
    c->pool.emplace_back(v->bytes, v->base);
    if (!arena_slot) c->pool_bytes += v->bytes;
  } else {
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(v->base);
  }
  delete v;
  return STORM_HIP_OK;
}

int storm_hip_vec_size(const storm_hip_vec *v, int64_t *n_owned, int64_t *n_halo) {
  STORM_REQUIRE(v, "vec_size: null vector");
  if (n_owned) *n_owned = v->n_owned;
  if (n_halo) *n_halo = v->n_halo;
  return STORM_HIP_OK;
}

int storm_hip_vec_upload(storm_hip_vec *v, const double *host, int64_t n) {
  STORM_REQUIRE(v && (host || n == 0), "vec_upload: null argument");
  STORM_REQUIRE(n == v->n_owned, "vec_upload: %lld values for a vector of %lld owned rows",
                (long long)n, (long long)v->n_owned);
  STORM_TRY(lazy_sync(v->ctx));
  HIP_TRY(hipMemcpyAsync(v->d, host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, v->ctx->stream));
  HIP_TRY(hipStreamSynchronize(v->ctx->stream));
  return STORM_HIP_OK;
}

int storm_hip_vec_download(const storm_hip_vec *v, double *host, int64_t n) {
  STORM_REQUIRE(v && (host || n == 0), "vec_download: null argument");
  STORM_REQUIRE(n == v->n_owned || n == v->n_owned + v->n_halo,
                "vec_download: %lld values requested from a vector of %lld(+%lld) rows", (long long)n,
                (long long)v->n_owned, (long long)v->n_halo);
  STORM_TRY(lazy_sync(v->ctx));
  HIP_TRY(hipMemcpyAsync(host, v->d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, v->ctx->stream));
  HIP_TRY(hipStreamSynchronize(v->ctx->stream));
  return STORM_HIP_OK;
}

static std::mt19937_64 &random_engine() {
  static std::mt19937_64 engine{};  // MatrixAlgorithms.hpp:145
  return engine;
}

void storm_hip_rng_reset(void) { random_engine() = std::mt19937_64{}; }

int storm_hip_fill_randomly(storm_hip_vec *v) {
  STORM_REQUIRE(v, "fill_randomly: null vector");
  STORM_REQUIRE(v->ctx->n_ranks == 1, "fill_randomly: the sequential generator is defined for a single rank only");
  std::uniform_real_distribution<double> distribution{0.0, 1.0};  // :146
  std::vector<double> host((size_t)v->n_owned);
  for (double &value : host) value = distribution(random_engine());
  return storm_hip_vec_upload(v, host.data(), v->n_owned);
}

int storm_hip_vec_device_ptr(storm_hip_vec *v, void **dev_ptr) {
  STORM_REQUIRE(v && dev_ptr, "vec_device_ptr: null argument");
  STORM_TRY(lazy_sync(v->ctx));  // (the caller is about to touch the memory itself)
  v->exposed = true;             // (... and may keep the address: this vector's storage stays where it is)
  *dev_ptr = v->d;
  return STORM_HIP_OK;
}

int storm_hip_vec_context(const storm_hip_vec *v, storm_hip_ctx **ctx) {
  STORM_REQUIRE(v && ctx, "vec_context: null argument");
  *ctx = v->ctx;
  return STORM_HIP_OK;
}

int storm_hip_vec_get(const storm_hip_vec *v, int64_t row, double *value) {
  STORM_REQUIRE(v && value, "vec_get: null argument");
  STORM_REQUIRE(row >= 0 && row < v->n_owned + v->n_halo, "vec_get: row %lld outside [0, %lld)", (long long)row,
                (long long)(v->n_owned + v->n_halo));
  HIP_TRY(hipSetDevice(v->ctx->device));
  STORM_TRY(lazy_sync(v->ctx));
  HIP_TRY(hipMemcpyAsync(value, v->d + row, sizeof(double), hipMemcpyDeviceToHost, v->ctx->stream));
  HIP_TRY(hipStreamSynchronize(v->ctx->stream));
  return STORM_HIP_OK;
}

}  // extern "C"
