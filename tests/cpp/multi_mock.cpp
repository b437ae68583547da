// multi_mock.cpp — stm::MultiDeviceBDPT (stratum_amd/host/stratum_hip_multi.hpp) at world sizes 2..8 WITHOUT GPUs: the C
// ABI, the HIP runtime calls and the RCCL calls the driver makes are replaced by host stand-ins defined in this file
// ("device memory" is host memory, streams and events complete at once, ncclSend / ncclRecv move bytes between the rank
// threads through mailboxes and block like the real ones when a peer never posts). What runs unmodified is the driver: its
// persistent rank threads, the two phases of a frame, the packing and assembly through ShardLayout, the frames in flight of
// the pipelined mode, the scene update taking the same path on every rank, and — the reason this test exists — what happens
// when one rank fails: the call must throw before a single collective has been posted, and the driver must stay usable.
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
int sthip_render(sthip_ctx* ctx, const sthip_BDPTPushConstants* pc, uint32_t, uint32_t, const sthip_frame_desc*, uint32_t seed_begin, uint32_t, const sthip_outputs* o) {
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
  for (uint32_t s = 0; s < n; s++) {
    uint32_t x, y;
    const bool inside = L.slot_pixel(c->rank, s, x, y);
    for (int k = 0; k < 4; k++) rad[4 * (size_t)s + k] = inside ? mock::radiance_of(x, y, seed_begin, k) : 0.f;
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
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  std::memcpy(d, s, n);
  return hipSuccess;
}
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
  {
    std::lock_guard<std::mutex> lk(mock::mail_mutex);
    mock::mail[{me, peer}].emplace_back((const char*)buf, (const char*)buf + count);
  }
  mock::mail_cv.notify_all();
  return ncclSuccess;
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t) {
  mock::nccl_calls++;
  const int me = reinterpret_cast<MockComm*>(comm)->rank;
  std::unique_lock<std::mutex> lk(mock::mail_mutex);
  auto& q = mock::mail[{peer, me}];
  mock::mail_cv.wait(lk, [&]() { return !q.empty(); });  // (a peer that never sends: main()'s alarm ends the test)
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
    std::printf("MULTI MOCK OK world %d frame %ux%u tiles %ux%u collectives %d\n", world, L.W, L.H, tw, th, mock::nccl_calls.load());
    return 0;
  } catch (const std::exception& e) {
    std::printf("ERROR: %s\n", e.what());
    return 1;
  }
}
