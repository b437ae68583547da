// multi_mock.cpp — stm::MultiDeviceBDPT (stratum_amd/host/stratum_hip_multi.hpp) at world sizes 2..8 WITHOUT GPUs: the C
// ABI, the HIP runtime calls and the RCCL calls the driver makes are replaced by host stand-ins defined in this file
// ("device memory" is host memory, streams and events complete at once, ncclSend / ncclRecv move bytes between the rank
// threads through mailboxes and block like the real ones when a peer never posts). What runs unmodified is the driver: its
// persistent rank threads, the two phases of a frame, the packing and assembly through ShardLayout, the frames in flight of
// the pipelined mode, the scene update taking the same path on every rank, and — the reason this test exists — what happens
// when one rank fails: the call must throw before a single collective has been posted, and the driver must stay usable; a
// failure INSIDE the exchange (an ncclSend that errors on one rank, an asynchronous error while rank 0 waits) must abort every
// communicator so that no peer is left waiting, throw, and leave a driver that renders the next frame. Also the seed-split
// replica mode (whole frames over disjoint seed ranges + one ncclReduce(sum)) and read_back(false).
//   multi_mock <scene.bin> <world> [tile_w tile_h]
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>

#include "../../stratum_amd/host/stratum_hip_multi.hpp"
#include "scene_reader.hpp"

using namespace stm;

// ------------------------------------------------------------------ the stand-ins ------------------------------------------------------------------
namespace mock {
struct Ctx {
  int device = 0;
  uint32_t rank = 0, world = 1, tile_w = 64, tile_h = 32;
  std::string error;
  int uploads = 0, updates = 0;
};
std::atomic<int> fail_render_rank{-1};   // sthip_render fails on this rank
std::atomic<int> fail_update_rank{-1};   // sthip_scene_update_transforms fails on this rank
std::atomic<int> nccl_calls{0};          // ncclSend + ncclRecv calls so far
std::atomic<int> fail_send_rank{-1};     // ncclSend / ncclReduce fails on this rank (phase 2)
std::atomic<int> async_error_rank{-1};   // ncclCommGetAsyncError reports an error on this rank's communicator
std::atomic<bool> stall_events{false};   // hipEventQuery says "not ready" (an exchange that never completes)
std::atomic<int> aborts{0};              // ncclCommAbort calls so far
std::atomic<int> comm_inits{0};
std::atomic<int> d2h_copies{0};          // device-to-host copies of more than 16 bytes (frames; the ray counts are 16)
bool aborted = false;                    // (under mail_mutex) the communicators of this generation were aborted
int reduce_world = 0;                    // ranks of the communicators (ncclCommInitAll)
std::mutex mail_mutex;
std::condition_variable mail_cv;
std::map<std::pair<int, int>, std::deque<std::vector<char>>> mail;  // (from, to) -> messages in order

uint32_t slot_count(uint32_t w, uint32_t h, uint32_t rank, uint32_t world, uint32_t tw, uint32_t th) {
  if (!w || !h || !world || !tw || !th || rank >= world) return 0;
  const uint32_t tiles = ((w + tw - 1) / tw) * ((h + th - 1) / th);
  const uint32_t owned = tiles > rank ? (tiles - rank + world - 1) / world : 0;
  return owned * tw * th;
}
// what the stand-in renderer writes: a value per (pixel, seed, output), so that every misplaced entry shows
float radiance_of(uint32_t x, uint32_t y, uint32_t seed, int c) { return (float)((x * 7919u + y * 104729u + seed * 13u + (uint32_t)c) % 65521u) + 0.5f; }
uint32_t aov_word(uint32_t x, uint32_t y, int output, int word) { return (x * 31u + y * 17u) * 8u + (uint32_t)output * 4u + (uint32_t)word + 1u; }
}  // namespace mock

extern "C" {
int sthip_create(int device, sthip_ctx** out) {
  auto* c = new mock::Ctx();
  c->device = device;
  *out = reinterpret_cast<sthip_ctx*>(c);
  return STHIP_OK;
}
void sthip_destroy(sthip_ctx* ctx) { delete reinterpret_cast<mock::Ctx*>(ctx); }
const char* sthip_last_error(const sthip_ctx* ctx) { return ctx ? reinterpret_cast<const mock::Ctx*>(ctx)->error.c_str() : "no context"; }
int sthip_set_stream(sthip_ctx*, void*) { return STHIP_OK; }
int sthip_set_option(sthip_ctx*, const char*, int64_t) { return STHIP_OK; }
int sthip_set_shard(sthip_ctx* ctx, uint32_t rank, uint32_t count, uint32_t tw, uint32_t th) {
  auto* c = reinterpret_cast<mock::Ctx*>(ctx);
  c->rank = rank;
  c->world = count;
  c->tile_w = tw;
  c->tile_h = th;
  return STHIP_OK;
}
uint32_t sthip_shard_slot_count(uint32_t w, uint32_t h, uint32_t rank, uint32_t world, uint32_t tw, uint32_t th) { return mock::slot_count(w, h, rank, world, tw, th); }
int sthip_scene_upload(sthip_ctx* ctx, const sthip_scene_desc*) {
  reinterpret_cast<mock::Ctx*>(ctx)->uploads++;
  return STHIP_OK;
}
int sthip_scene_update_transforms(sthip_ctx* ctx, const sthip_TransformData*, const sthip_TransformData*, const sthip_TransformData*, uint32_t) {
  auto* c = reinterpret_cast<mock::Ctx*>(ctx);
  if (mock::fail_update_rank.load() == (int)c->rank && c->rank != 0) {
    c->error = "stand-in: the update is refused on this rank";
    return STHIP_ERR_UNSUPPORTED;
  }
  c->updates++;
  return STHIP_OK;
}
int sthip_radiance_to_sums(sthip_ctx*, float* image, uint64_t entries, uint32_t back) {
  for (uint64_t i = 0; i < entries; i++) {
    float* v = image + 4 * i;
    for (int c = 0; c < 3; c++) {
      if (!back) v[c] = v[c] * v[3];
      else if (v[3] > 0.0f) v[c] = v[c] / v[3];
    }
  }
  return STHIP_OK;
}
int sthip_render(sthip_ctx* ctx, const sthip_BDPTPushConstants* pc, uint32_t, uint32_t, const sthip_frame_desc*, uint32_t seed_begin, uint32_t seed_count, const sthip_outputs* o) {
  auto* c = reinterpret_cast<mock::Ctx*>(ctx);
  if (mock::fail_render_rank.load() == (int)c->rank) {
    c->error = "stand-in: this rank fails";
    return STHIP_ERR_HIP;
  }
  const uint32_t W = pc->gOutputExtent[0], H = pc->gOutputExtent[1];
  const ShardLayout L{W, H, c->world, c->tile_w, c->tile_h};
  const uint32_t n = L.slot_count(c->rank);
  float* rad = o->gRadiance;
  uint64_t owned = 0;
  if (o->gAlbedo) std::memset(o->gAlbedo, 0, (size_t)W * H * 16);
  if (o->gVisibility) std::memset(o->gVisibility, 0, (size_t)W * H * 8);
  if (o->gDepth) std::memset(o->gDepth, 0, (size_t)W * H * 16);
  if (o->gPrevUVs) std::memset(o->gPrevUVs, 0, (size_t)W * H * 8);
  const bool image = o->radiance_layout == STHIP_LAYOUT_IMAGE;  // (the whole frame: the seed-split mode; rgb = mean over the call's seeds, a = their number)
  for (uint32_t s = 0; s < n; s++) {
    uint32_t x, y;
    const bool inside = L.slot_pixel(c->rank, s, x, y);
    if (image) {
      if (inside) {
        float* v = rad + 4 * ((size_t)y * W + x);
        for (int k = 0; k < 3; k++) {
          float sum = 0.f;
          for (uint32_t q = 0; q < seed_count; q++) sum += mock::radiance_of(x, y, seed_begin + q, k);
          v[k] = sum / (float)seed_count;
        }
        v[3] = (float)seed_count;
      }
    } else {
      for (int k = 0; k < 4; k++) rad[4 * (size_t)s + k] = inside ? mock::radiance_of(x, y, seed_begin, k) : 0.f;
    }
    if (!inside) continue;
    owned++;
    const size_t px = (size_t)y * W + x;
    uint32_t* out[4] = {(uint32_t*)o->gAlbedo, (uint32_t*)o->gVisibility, (uint32_t*)o->gDepth, (uint32_t*)o->gPrevUVs};
    const int words[4] = {4, 2, 4, 2};
    for (int a = 0; a < 4; a++)
      if (out[a])
        for (int k = 0; k < words[a]; k++) out[a][px * words[a] + k] = mock::aov_word(x, y, a, k);
  }
  if (o->gRayCount) {
    o->gRayCount[0] = 5 * owned;
    o->gRayCount[1] = 3 * owned;
  }
  return STHIP_OK;
}
int sthip_pack_tiles(sthip_ctx* ctx, const void* image, uint32_t W, uint32_t H, uint32_t bytes, void* packed) {
  auto* c = reinterpret_cast<mock::Ctx*>(ctx);
  const ShardLayout L{W, H, c->world, c->tile_w, c->tile_h};
  for (uint32_t s = 0, n = L.slot_count(c->rank); s < n; s++) {
    uint32_t x, y;
    if (L.slot_pixel(c->rank, s, x, y))
      std::memcpy((char*)packed + (size_t)s * bytes, (const char*)image + ((size_t)y * W + x) * bytes, bytes);
    else
      std::memset((char*)packed + (size_t)s * bytes, 0, bytes);
  }
  return STHIP_OK;
}
int sthip_assemble_tiles_bytes(sthip_ctx*, const void* packed, uint64_t stride, uint32_t world, uint32_t tw, uint32_t th, uint32_t W, uint32_t H, uint32_t bytes, void* frame) {
  const ShardLayout L{W, H, world, tw, th};
  std::vector<const void*> parts;
  for (uint32_t r = 0; r < world; r++) parts.push_back((const char*)packed + (size_t)r * stride * bytes);
  L.assemble(parts, frame, bytes);
  return STHIP_OK;
}
int sthip_tonemap(sthip_ctx*, const sthip_tonemap_desc* d) {
  if (d->gOutput && d->gInput) std::memcpy(d->gOutput, d->gInput, (size_t)d->width * d->height * 16);
  return STHIP_OK;
}
int sthip_write_hdr(const char*, uint32_t, uint32_t, const float*) { return STHIP_OK; }

// ---- HIP runtime: host memory, everything completes at once ----
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) {
  *p = std::malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipMemset(void* p, int v, size_t n) {
  std::memset(p, v, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind kind, hipStream_t) {
  if (kind == hipMemcpyDeviceToHost && n > 16) mock::d2h_copies++;
  std::memcpy(d, s, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  std::memset(p, v, n);
  return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t) { return mock::stall_events.load() ? hipErrorNotReady : hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = reinterpret_cast<hipStream_t>(new int(0));
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  delete reinterpret_cast<int*>(s);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
  *e = reinterpret_cast<hipEvent_t>(new int(0));
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  delete reinterpret_cast<int*>(e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stand-in HIP error"; }

// ---- RCCL: point-to-point through mailboxes; a receive blocks until its peer has sent (for ever, like the real one) ----
struct MockComm {
  int rank;
};
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) {
  for (int r = 0; r < n; r++) comms[r] = reinterpret_cast<ncclComm_t>(new MockComm{r});
  std::lock_guard<std::mutex> lk(mock::mail_mutex);
  mock::mail.clear();  // a new generation: what the aborted one left in flight is gone
  mock::reduce_world = n;
  mock::aborted = false;
  mock::comm_inits++;
  return ncclSuccess;
}
// ends what its peers are blocked in (the stand-in keeps the object until the process ends: a peer may still be inside a call on it)
static std::vector<std::unique_ptr<MockComm>> graveyard;
ncclResult_t ncclCommAbort(ncclComm_t c) {
  mock::aborts++;
  {
    std::lock_guard<std::mutex> lk(mock::mail_mutex);
    mock::aborted = true;
    graveyard.emplace_back(reinterpret_cast<MockComm*>(c));
  }
  mock::mail_cv.notify_all();
  return ncclSuccess;
}
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* e) {
  *e = mock::async_error_rank.load() == reinterpret_cast<MockComm*>(comm)->rank ? ncclSystemError : ncclSuccess;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  delete reinterpret_cast<MockComm*>(c);
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t) { return "stand-in RCCL error"; }
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t) {
  mock::nccl_calls++;
  const int me = reinterpret_cast<MockComm*>(comm)->rank;
  if (mock::fail_send_rank.load() == me) return ncclSystemError;
  {
    std::lock_guard<std::mutex> lk(mock::mail_mutex);
    if (mock::aborted) return ncclInternalError;
    mock::mail[{me, peer}].emplace_back((const char*)buf, (const char*)buf + count);
  }
  mock::mail_cv.notify_all();
  return ncclSuccess;
}
// sum-reduce of floats to `root`: the others send, the root adds in rank order (a fixed order: the check below repeats it)
ncclResult_t ncclReduce(const void* send, void* recv, size_t count, ncclDataType_t, ncclRedOp_t, int root, ncclComm_t comm, hipStream_t) {
  mock::nccl_calls++;
  const int me = reinterpret_cast<MockComm*>(comm)->rank;
  if (mock::fail_send_rank.load() == me) return ncclSystemError;
  if (me != root) {
    {
      std::lock_guard<std::mutex> lk(mock::mail_mutex);
      if (mock::aborted) return ncclInternalError;
      mock::mail[{me, root}].emplace_back((const char*)send, (const char*)send + 4 * count);
    }
    mock::mail_cv.notify_all();
    return ncclSuccess;
  }
  std::vector<float> acc((const float*)send, (const float*)send + count);
  int world = 0;
  for (;; world++) {  // (the stand-in learns the world size from who has a mailbox towards the root: every rank > 0 posts one)
    if (world == me) continue;
    std::unique_lock<std::mutex> lk(mock::mail_mutex);
    if (world >= mock::reduce_world) break;
    auto& q = mock::mail[{world, me}];
    mock::mail_cv.wait(lk, [&]() { return !q.empty() || mock::aborted; });
    if (mock::aborted) return ncclInternalError;
    const float* v = (const float*)q.front().data();
    for (size_t i = 0; i < count; i++) acc[i] += v[i];
    q.pop_front();
  }
  std::memcpy(recv, acc.data(), 4 * count);
  return ncclSuccess;
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t) {
  mock::nccl_calls++;
  const int me = reinterpret_cast<MockComm*>(comm)->rank;
  std::unique_lock<std::mutex> lk(mock::mail_mutex);
  auto& q = mock::mail[{peer, me}];
  mock::mail_cv.wait(lk, [&]() { return !q.empty() || mock::aborted; });  // (a peer that never sends and nobody aborts: main()'s alarm ends the test)
  if (mock::aborted) return ncclInternalError;
  if (q.front().size() != count) return ncclInvalidArgument;
  std::memcpy(buf, q.front().data(), count);
  q.pop_front();
  return ncclSuccess;
}
}  // extern "C"

// ------------------------------------------------------------------ the test ------------------------------------------------------------------
static int check_frame(const BDPT::Frame& fr, uint32_t W, uint32_t H, uint32_t seed, bool aovs, const char* what) {
  if (fr.width != W || fr.height != H || fr.mRadiance.size() != 4 * (size_t)W * H) return std::printf("FAIL (%s): frame size\n", what), 1;
  for (uint32_t y = 0; y < H; y++)
    for (uint32_t x = 0; x < W; x++) {
      const size_t px = (size_t)y * W + x;
      for (int k = 0; k < 4; k++)
        if (fr.mRadiance[4 * px + k] != mock::radiance_of(x, y, seed, k)) return std::printf("FAIL (%s): radiance at %u,%u is %g\n", what, x, y, fr.mRadiance[4 * px + k]), 1;
      if (!aovs) continue;
      const uint32_t* a[4] = {(const uint32_t*)fr.mAlbedo.data(), (const uint32_t*)fr.mVisibility.data(), (const uint32_t*)fr.mDepth.data(), (const uint32_t*)fr.mPrevUVs.data()};
      const int words[4] = {4, 2, 4, 2};
      for (int o = 0; o < 4; o++)
        for (int k = 0; k < words[o]; k++)
          if (a[o][px * words[o] + k] != mock::aov_word(x, y, o, k)) return std::printf("FAIL (%s): G-buffer output %d at %u,%u\n", what, o, x, y), 1;
    }
  if (fr.mRayCount[0] != 5ull * W * H || fr.mRayCount[1] != 3ull * W * H) return std::printf("FAIL (%s): ray counts %llu %llu\n", what, (unsigned long long)fr.mRayCount[0], (unsigned long long)fr.mRayCount[1]), 1;
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return std::fprintf(stderr, "usage: multi_mock scene.bin world [tile_w tile_h]\n"), 2;
  alarm(240);  // a rank left waiting for a peer is a hang: SIGALRM ends the process, the test fails
  try {
    const int world = atoi(argv[2]);
    const uint32_t tw = argc > 4 ? (uint32_t)atoi(argv[3]) : 64, th = argc > 4 ? (uint32_t)atoi(argv[4]) : 32;
    std::vector<int> devices;
    for (int r = 0; r < world; r++) devices.push_back(r);
    Reader rd(argv[1]);
    NodeGraph graph;
    Node& root = graph.emplace("Instance");
    auto app = root.make_child("Application").make_component<Application>();
    LoadedScene L = load_scene(rd, app.node());
    auto renderer = app.node().make_child("BDPT").make_component<MultiDeviceBDPT>(devices, tw, th);
    uint32_t seeds = 1;
    app->OnRenderWindow.add_listener(renderer.node(), [&](CommandBuffer& c) { renderer->render(c, L.W, L.H, {{L.view, L.view_xf}}, seeds); });
    CommandBuffer cb;
    // 1. one frame, every output exchanged
    app->run_frame(cb);
    if (check_frame(renderer->prev_result(), L.W, L.H, 0, true, "first frame")) return 1;
    // 2. radiance only
    renderer->gather_aovs(false);
    app->run_frame(cb);
    if (check_frame(renderer->prev_result(), L.W, L.H, 1, false, "radiance only")) return 1;
    renderer->gather_aovs(true);
    // 3. a failing rank: the call throws, no collective has been posted, nobody waits; afterwards the driver works again
    for (int bad = 0; bad < world; bad += world > 2 ? world - 1 : 1) {
      const int before = mock::nccl_calls.load();
      mock::fail_render_rank = bad;
      bool threw = false;
      try {
        app->run_frame(cb);
      } catch (const std::exception& e) {
        threw = std::string(e.what()).find("this rank fails") != std::string::npos;
      }
      mock::fail_render_rank = -1;
      if (!threw) return std::printf("FAIL: a failing rank %d did not surface as an exception\n", bad), 1;
      if (mock::nccl_calls.load() != before) return std::printf("FAIL: collectives were posted although rank %d had failed\n", bad), 1;
      app->run_frame(cb);  // (the failed call consumed no seed: the frame number moves on with completed submissions only)
      if (check_frame(renderer->prev_result(), L.W, L.H, 2 + (bad ? 1 : 0), true, "after a failure")) return 1;
    }
    // 4. frames in flight: prev_result lags one call behind until flush()
    const uint32_t base = world > 2 ? 4 : 4;
    renderer->pipelined(true);
    app->run_frame(cb);  // submits `base`
    app->run_frame(cb);  // submits base + 1, completes base
    if (check_frame(renderer->prev_result(), L.W, L.H, base, true, "pipelined, one behind")) return 1;
    app->run_frame(cb);
    if (check_frame(renderer->prev_result(), L.W, L.H, base + 1, true, "pipelined, one behind (2)")) return 1;
    renderer->flush();
    if (check_frame(renderer->prev_result(), L.W, L.H, base + 2, true, "pipelined, flushed")) return 1;
    renderer->pipelined(false);
    // 5. several seeds per call move the frame number by as many
    seeds = 3;
    app->run_frame(cb);
    if (check_frame(renderer->prev_result(), L.W, L.H, base + 3, true, "three seeds")) return 1;
    seeds = 1;
    app->run_frame(cb);
    if (check_frame(renderer->prev_result(), L.W, L.H, base + 6, true, "after three seeds")) return 1;
    // 6. a failure INSIDE the exchange: ncclSend errors on one rank while rank 0 already waits in its receives. The failing
    // thread aborts every communicator, the call throws (it must not hang: the alarm would end the test), the next call
    // makes new communicators and renders
    uint32_t next_seed = base + 7;
    for (int bad : {world - 1, world > 3 ? 3 : 0}) {
      const int aborts_before = mock::aborts.load(), inits_before = mock::comm_inits.load();
      mock::fail_send_rank = bad;
      bool threw = false;
      try {
        app->run_frame(cb);
      } catch (const std::exception& e) {
        threw = true;
      }
      mock::fail_send_rank = -1;
      if (!threw) return std::printf("FAIL: a failing ncclSend on rank %d did not surface as an exception\n", bad), 1;
      if (mock::aborts.load() < aborts_before + world) return std::printf("FAIL: %d of %d communicators were aborted after rank %d failed in the exchange\n", mock::aborts.load() - aborts_before, world, bad), 1;
      app->run_frame(cb);
      if (mock::comm_inits.load() != inits_before + 1) return std::printf("FAIL: the communicators were not made again after an abort\n"), 1;
      // (the failed call consumed no seed: the frame number moves on with completed submissions only)
      if (check_frame(renderer->prev_result(), L.W, L.H, next_seed++, true, "after a failed exchange")) return 1;
    }
    // 7. an asynchronous error while rank 0 waits for an exchange that never completes: finish() polls, aborts, throws
    {
      const int aborts_before = mock::aborts.load();
      mock::stall_events = true;
      mock::async_error_rank = world - 1;
      bool threw = false;
      try {
        app->run_frame(cb);
      } catch (const std::exception& e) {
        threw = std::string(e.what()).find("asynchronous error") != std::string::npos;
      }
      mock::stall_events = false;
      mock::async_error_rank = -1;
      if (!threw) return std::printf("FAIL: an asynchronous RCCL error did not end the wait for the exchange\n"), 1;
      if (mock::aborts.load() < aborts_before + world) return std::printf("FAIL: communicators not aborted after an asynchronous error\n"), 1;
      next_seed++;  // (that frame had been submitted — rendered and its exchange posted — when the wait for it failed: its seed is spent)
      app->run_frame(cb);
      if (check_frame(renderer->prev_result(), L.W, L.H, next_seed++, true, "after an asynchronous error")) return 1;
    }
    // 8. read_back(false): the frame stays on rank 0's device, no image crosses to the host
    {
      renderer->read_back(false);
      const int copies = mock::d2h_copies.load();
      app->run_frame(cb);
      if (mock::d2h_copies.load() != copies) return std::printf("FAIL: read_back(false) still copied %d images to the host\n", mock::d2h_copies.load() - copies), 1;
      const BDPT::Frame& fr = renderer->prev_result();
      if (!fr.mRadiance.empty() || fr.mRayCount[0] != 5ull * L.W * L.H) return std::printf("FAIL: read_back(false): host frame / ray counts\n"), 1;
      const float* dev = renderer->device_frame();  // ("device" memory is host memory here)
      for (uint32_t y = 0; y < L.H; y += 7)
        for (uint32_t x = 0; x < L.W; x += 5)
          for (int k = 0; k < 4; k++)
            if (dev[4 * ((size_t)y * L.W + x) + k] != mock::radiance_of(x, y, next_seed, k)) return std::printf("FAIL: read_back(false): device frame at %u,%u\n", x, y), 1;
      const uint32_t* alb = (const uint32_t*)renderer->device_albedo();
      if (alb[4 * 3 + 1] != mock::aov_word(3, 0, 0, 1)) return std::printf("FAIL: read_back(false): device albedo\n"), 1;
      next_seed++;
      renderer->read_back(true);
    }
    // 9. the seed-split replica mode: whole frames over disjoint seed ranges, one sum-reduce; more ranks than seeds; pipelined
    {
      renderer->split_seeds(true);
      auto expect = [&](uint32_t first, uint32_t count, const char* what) {
        const BDPT::Frame& fr = renderer->prev_result();
        if (fr.width != L.W || fr.mRadiance.size() != 4 * (size_t)L.W * L.H) return std::printf("FAIL (%s): frame size\n", what), 1;
        for (uint32_t y = 0; y < L.H; y++)
          for (uint32_t x = 0; x < L.W; x++) {
            float want[4] = {0, 0, 0, 0};
            for (int r = 0; r < world; r++) {  // what each rank renders, turns into sums, and the reduce adds in rank order
              const uint32_t s0 = (count / world) * r + std::min<uint32_t>(r, count % world), s1 = (count / world) * (r + 1) + std::min<uint32_t>(r + 1, count % world);
              if (s1 == s0) continue;
              for (int k = 0; k < 3; k++) {
                float sum = 0.f;
                for (uint32_t q = s0; q < s1; q++) sum += mock::radiance_of(x, y, first + q, k);
                want[k] += (sum / (float)(s1 - s0)) * (float)(s1 - s0);
              }
              want[3] += (float)(s1 - s0);
            }
            for (int k = 0; k < 3; k++) want[k] = want[k] / want[3];
            for (int k = 0; k < 4; k++)
              if (fr.mRadiance[4 * ((size_t)y * L.W + x) + k] != want[k]) return std::printf("FAIL (%s): radiance at %u,%u channel %d is %g, expected %g\n", what, x, y, k, fr.mRadiance[4 * ((size_t)y * L.W + x) + k], want[k]), 1;
            const uint32_t* a = (const uint32_t*)fr.mAlbedo.data();
            if (a[4 * ((size_t)y * L.W + x) + 2] != mock::aov_word(x, y, 0, 2)) return std::printf("FAIL (%s): albedo\n", what), 1;
          }
        if (fr.mRayCount[0] != 5ull * L.W * L.H * std::min<uint32_t>(count, world)) return std::printf("FAIL (%s): ray counts %llu\n", what, (unsigned long long)fr.mRayCount[0]), 1;
        return 0;
      };
      seeds = 2 * (uint32_t)world + 1;  // uneven ranges
      app->run_frame(cb);
      if (expect(next_seed, seeds, "seed split, uneven")) return 1;
      next_seed += seeds;
      seeds = 1;  // more ranks than seeds: the others add zeros
      app->run_frame(cb);
      if (expect(next_seed, 1, "seed split, one seed")) return 1;
      next_seed += 1;
      seeds = (uint32_t)world;
      renderer->pipelined(true);
      app->run_frame(cb);
      app->run_frame(cb);
      if (expect(next_seed, seeds, "seed split, pipelined")) return 1;
      renderer->flush();
      if (expect(next_seed + seeds, seeds, "seed split, flushed")) return 1;
      next_seed += 2 * seeds;
      renderer->pipelined(false);
      renderer->split_seeds(false);
      seeds = 1;
      app->run_frame(cb);
      if (check_frame(renderer->prev_result(), L.W, L.H, next_seed++, true, "tiles again")) return 1;
    }
    std::printf("MULTI MOCK OK world %d frame %ux%u tiles %ux%u collectives %d\n", world, L.W, L.H, tw, th, mock::nccl_calls.load());
    return 0;
  } catch (const std::exception& e) {
    std::printf("ERROR: %s\n", e.what());
    return 1;
  }
}
