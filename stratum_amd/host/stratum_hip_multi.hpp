// stratum_hip_multi.hpp — the multi-GPU driver of the C++ host side: one process, one sthip context per GPU, one
// persistent host thread per context, the frame cut into pixel tiles (sthip_set_shard), and ONE exchange per render call:
// an RCCL gather of the ranks' packed tiles to the first device over xGMI, scattered into the images there
// (sthip_assemble_tiles*).
//
// The reference has nothing to compare with: it creates one vk::Device with one queue per family and renders on it
// (src/Core/Device.cpp:125-131). This is the component BASELINE.json's north star adds ("frames shard by pixel-tile
// across the 8 GPUs of one node with an RCCL reduce of the accumulation buffer over xGMI") on the host language the
// north star keeps (C++). The path shards without any data-path communication: a pixel's RNG key is
// (x, y, seed, counter) (rng.hlsli:35-47) and a pixel owns its outputs, so the ranks only meet to assemble the frame.
//
// stm::MultiDeviceBDPT is a stm::BDPT (same component, same update()/render() the Application drives, BDPT.hpp:23-24):
// its base context renders rank 0, the extra contexts the other ranks. A frame is made in two phases:
//   1. every rank renders its tiles on its own stream (worker threads, in parallel). If ANY rank fails, the call throws here:
//      not a single collective has been issued, so nothing is left waiting for a peer that will never come;
//   2. every rank posts its part of the exchange — ncclSend of its packed tiles, rank 0 the matching ncclRecvs, one group —
//      on its communication stream, behind the render (an event, no host wait). The radiance always travels; the G-buffer
//      outputs the reference's render always produces (albedo, VisibilityInfo, DepthInfo, prev-uv: bdpt.hlsl:222-296, what
//      the denoiser consumes, Denoiser.cpp:186-213) travel the same way when gather_aovs(true) (the default).
// With pipelined(true) the exchange, the assembly and the read-back of frame i run while frame i + 1 renders (two sets of
// buffers; the assembly and the copies go to rank 0's COMMUNICATION stream, so they do not queue behind that render):
// render() then returns once frame i - 1 is complete, prev_result() lags one call behind, flush() completes the last frame.
//
// split_seeds(true) is the other way to spread a call ("replicas + sum-reduce"): every GPU renders the WHOLE
// frame for its own part of the call's seed range, the images become sums (sthip_radiance_to_sums) and ONE
// ncclReduce(sum) of the accumulation buffer adds them on rank 0. It is what the whole-frame estimators need (light tracing,
// reservoir reuse: a tile shard is refused for them); reservoir reuse then restarts its chain of grids at the first seed of
// every rank's range. The G-buffer is the same on every rank: rank 0's is used.
//
// A failure inside the exchange (phase 2) cannot leave peers waiting: the thread that fails aborts EVERY communicator
// (ncclCommAbort), which ends the calls its peers are blocked in; the call throws, the communicators are made again at the
// next render. finish() polls ncclCommGetAsyncError while it waits for an exchange to complete, with the same outcome.
// read_back(false) leaves the assembled frame on devices[0] (device_frame() and friends): no device-to-host copies at all.
// Needs <rccl/rccl.h> and the HIP runtime (link -lrccl -lamdhip64); stratum_hip.hpp itself stays free of both.
#pragma once

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "stratum_hip.hpp"

namespace stm {

// ---- the host mirror of the tile ownership rule and of the packed layout (kernels.h: slot_to_pixel) ----
struct ShardLayout {
  uint32_t width, height, world, tile_w, tile_h;
  uint32_t tiles_x() const { return (width + tile_w - 1) / tile_w; }
  uint32_t tiles_y() const { return (height + tile_h - 1) / tile_h; }
  uint32_t owner_of_pixel(uint32_t x, uint32_t y) const { return ((y / tile_h) * tiles_x() + x / tile_w) % world; }  // tile t belongs to rank t % world
  uint32_t slot_count(uint32_t rank) const { return sthip_shard_slot_count(width, height, rank, world, tile_w, tile_h); }
  // pixel of slot `slot` of rank `rank`'s packed buffer; false for slots outside the image (padding of edge tiles)
  bool slot_pixel(uint32_t rank, uint32_t slot, uint32_t& px, uint32_t& py) const {
    const uint32_t per_tile = tile_w * tile_h, local_tile = slot / per_tile, r = slot % per_tile;
    const uint32_t tile = local_tile * world + rank;
    const uint32_t ty = tile / tiles_x(), tx = tile % tiles_x();
    const uint32_t b = r >> 6, lane = r & 63u, blocks_x = tile_w >> 3;
    px = tx * tile_w + ((b % blocks_x) << 3) + (lane & 7u);
    py = ty * tile_h + ((b / blocks_x) << 3) + (lane >> 3);
    return px < width && py < height && tile < tiles_x() * tiles_y();
  }
  // what sthip_assemble_tiles does on the device, on the host: packed[r] = rank r's buffer (slot_count(r) entries of
  // entry_bytes each)
  void assemble(const std::vector<const void*>& packed, void* frame, size_t entry_bytes = 16) const {
    std::memset(frame, 0, (size_t)width * height * entry_bytes);
    for (uint32_t r = 0; r < world; r++)
      for (uint32_t s = 0, n = slot_count(r); s < n; s++) {
        uint32_t x, y;
        if (slot_pixel(r, s, x, y)) std::memcpy((char*)frame + entry_bytes * ((size_t)y * width + x), (const char*)packed[r] + entry_bytes * (size_t)s, entry_bytes);
      }
  }
};

// One host thread per rank for the life of the driver (a context is single-threaded, different contexts may run
// concurrently: sthip.h). run(fn) hands fn(rank) to the threads of ranks [first, world) and returns when all are through;
// the first exception (by rank) is rethrown, after every thread has finished its job.
class RankThreads {
 public:
  explicit RankThreads(size_t world) : mJobs(world), mErrors(world) {
    for (size_t r = 0; r < world; r++) mThreads.emplace_back([this, r]() { loop(r); });
  }
  ~RankThreads() {
    {
      std::lock_guard<std::mutex> lk(mMutex);
      mStop = true;
    }
    mWake.notify_all();
    for (auto& t : mThreads) t.join();
  }
  void run(size_t first, const std::function<void(size_t)>& fn) {
    {
      std::lock_guard<std::mutex> lk(mMutex);
      for (size_t r = first; r < mJobs.size(); r++) {
        mJobs[r] = fn;
        mErrors[r].clear();
        mPending++;
      }
    }
    mWake.notify_all();
    std::unique_lock<std::mutex> lk(mMutex);
    mDone.wait(lk, [this]() { return mPending == 0; });
    for (size_t r = first; r < mJobs.size(); r++)
      if (!mErrors[r].empty()) throw std::runtime_error(mErrors[r]);
  }

 private:
  void loop(size_t r) {
    for (;;) {
      std::function<void(size_t)> job;
      {
        std::unique_lock<std::mutex> lk(mMutex);
        mWake.wait(lk, [this, r]() { return mStop || (bool)mJobs[r]; });
        if (mStop) return;
        job.swap(mJobs[r]);
      }
      std::string error;
      try {
        job(r);
      } catch (const std::exception& e) {
        error = e.what()[0] ? e.what() : "error";
      } catch (...) {
        error = "unknown error";
      }
      {
        std::lock_guard<std::mutex> lk(mMutex);
        mErrors[r] = error;
        mPending--;
      }
      mDone.notify_all();
    }
  }
  std::vector<std::thread> mThreads;
  std::vector<std::function<void(size_t)>> mJobs;
  std::vector<std::string> mErrors;
  std::mutex mMutex;
  std::condition_variable mWake, mDone;
  size_t mPending = 0;
  bool mStop = false;
};

class MultiDeviceBDPT : public BDPT {
 public:
  // devices[0] is rank 0 (the base class's context): it receives the tiles and holds the assembled frame
  explicit MultiDeviceBDPT(Node& node, const std::vector<int>& devices, uint32_t tile_w = 64, uint32_t tile_h = 32)
      : BDPT(node, devices.empty() ? 0 : devices[0]), mDevices(devices), mTileW(tile_w), mTileH(tile_h) {
    if (devices.empty()) throw std::invalid_argument("MultiDeviceBDPT: no devices");
    mRanks.resize(devices.size());
    mCommMutex = std::vector<std::mutex>(devices.size());
    mRanks[0].ctx = mCtx;
    for (size_t r = 1; r < devices.size(); r++)
      if (sthip_create(devices[r], &mRanks[r].ctx) != STHIP_OK) throw std::runtime_error(std::string("sthip_create: ") + sthip_last_error(nullptr));
    init_comms();
    for (size_t r = 0; r < devices.size(); r++) {
      check_hip(hipSetDevice(devices[r]), "hipSetDevice");
      check_hip(hipStreamCreateWithFlags(&mRanks[r].stream, hipStreamNonBlocking), "hipStreamCreate");
      check_hip(hipStreamCreateWithFlags(&mRanks[r].comm_stream, hipStreamNonBlocking), "hipStreamCreate");
      for (int k = 0; k < 2; k++) {
        check_hip(hipEventCreateWithFlags(&mRanks[r].rendered[k], hipEventDisableTiming), "hipEventCreate");
        check_hip(hipEventCreateWithFlags(&mRanks[r].exchanged[k], hipEventDisableTiming), "hipEventCreate");
      }
      (void)sthip_set_stream(mRanks[r].ctx, mRanks[r].stream);
    }
    mThreads.reset(new RankThreads(devices.size()));
  }
  ~MultiDeviceBDPT() override {
    mThreads.reset();
    for (size_t r = 0; r < mRanks.size(); r++) {
      (void)hipSetDevice(mDevices[r]);
      (void)hipStreamSynchronize(mRanks[r].stream);
      (void)hipStreamSynchronize(mRanks[r].comm_stream);
      if (mRanks[r].comm) (void)ncclCommDestroy(mRanks[r].comm);
      for (int k = 0; k < 2; k++) {
        for (void* p : mRanks[r].buf[k].all()) (void)hipFree(p);
        if (mRanks[r].rendered[k]) (void)hipEventDestroy(mRanks[r].rendered[k]);
        if (mRanks[r].exchanged[k]) (void)hipEventDestroy(mRanks[r].exchanged[k]);
      }
      if (r > 0 && mRanks[r].ctx) sthip_destroy(mRanks[r].ctx);  // rank 0's context is the base class's
      if (mRanks[r].stream) (void)hipStreamDestroy(mRanks[r].stream);
      if (mRanks[r].comm_stream) (void)hipStreamDestroy(mRanks[r].comm_stream);
    }
    (void)hipSetDevice(mDevices[0]);
    for (int k = 0; k < 2; k++)
      for (void* p : mGathered[k].all()) (void)hipFree(p);
    for (void* p : mFrameDev.all()) (void)hipFree(p);
  }
  size_t world() const { return mRanks.size(); }
  void gather_aovs(bool on) { mGatherAOVs = on; }  // also exchange albedo / visibility / depth / prev-uv (default on)
  void split_seeds(bool on) {                       // replicas of the whole frame over disjoint seed ranges + one sum-reduce, instead of tiles + a gather
    flush();
    mSplitSeeds = on;
  }
  void read_back(bool on) { mReadBack = on; }       // false: prev_result() carries the ray counts only, the images stay on devices[0]
  void pipelined(bool on) {
    flush();
    mPipelined = on;
  }

  // the scene goes to every GPU (replicated: a 1M-triangle scene is < 200 MB of 288 GB), in parallel. Every rank takes the
  // SAME path — a transforms-only update everywhere or a full upload everywhere — so all of them walk the same tree form.
  void update(CommandBuffer& cb, float dt) override {
    auto scene = mNode.find_in_ancestor<Scene>();
    if (!scene) scene = mNode.root().find_in_descendants<Scene>();
    if (!scene || !scene->data() || scene->data().get() == mBound) return;
    flush();
    BDPT::update(cb, dt);  // rank 0, and the bookkeeping (mBound, light count, environment)
    (void)sthip_set_stream(mCtx, mRanks[0].stream);
    const sthip_scene_desc d = scene->data()->desc();
    const bool transforms_only = last_update_was_transforms_only();
    std::vector<int> took_update(mRanks.size(), 0);
    mThreads->run(1, [&](size_t r) {
      if (transforms_only && sthip_scene_update_transforms(mRanks[r].ctx, d.gInstanceTransforms, d.gInstanceInverseTransforms, d.gInstanceMotionTransforms, d.instance_count) == STHIP_OK) {
        took_update[r] = 1;
        return;
      }
      if (sthip_scene_upload(mRanks[r].ctx, &d) != STHIP_OK) throw std::runtime_error(std::string("sthip_scene_upload (rank ") + std::to_string(r) + "): " + sthip_last_error(mRanks[r].ctx));
    });
    if (transforms_only)  // a rank that had to fall back to the full upload (cannot happen with identical histories) makes everyone do so
      for (size_t r = 1; r < mRanks.size(); r++)
        if (!took_update[r]) {
          mThreads->run(0, [&](size_t q) {
            if (sthip_scene_upload(mRanks[q].ctx, &d) != STHIP_OK) throw std::runtime_error(std::string("sthip_scene_upload (rank ") + std::to_string(q) + "): " + sthip_last_error(mRanks[q].ctx));
          });
          break;
        }
  }

  // One frame over all GPUs (see the header of this file). mPrevFrame holds the assembled outputs and the ray counts.
  void render(CommandBuffer&, uint32_t width, uint32_t height, const std::vector<std::pair<ViewData, TransformData>>& views, uint32_t seed_count = 1) override {
    FrameSetup fs;
    prepare_frame(width, height, views, fs);
    const ShardLayout layout{width, height, (uint32_t)mRanks.size(), mTileW, mTileH};
    const size_t stride = layout.slot_count(0);  // rank 0 owns the most tiles: equal-size messages
    const size_t pixels = (size_t)width * height;
    ensure_buffers(stride, pixels);
    const uint32_t seed_begin = frame_number();
    const int k = mPipelined ? (int)(mSubmitted & 1u) : 0;
    if (mInFlight[k].valid) finish(k);  // (cannot be: the frame that used these buffers was completed at the end of the call before last)
    if (mCommsAborted.load()) init_comms();    // an earlier exchange failed and took the communicators with it
    const bool seeds_mode = mSplitSeeds;
    const uint32_t world_n = (uint32_t)mRanks.size();
    // seeds mode: rank r takes seeds [first_of(r), first_of(r + 1)) of the call (the first seed_count % world ranks one more)
    auto first_of = [&](uint32_t r) { return (seed_count / world_n) * r + std::min(r, seed_count % world_n); };
    // ---- phase 1: every rank renders; no collective has been issued when this returns or throws ----
    mThreads->run(0, [&](size_t r) {
      Rank& rk = mRanks[r];
      Buffers& b = rk.buf[k];
      check_hip(hipSetDevice(mDevices[r]), "hipSetDevice");
      if (sthip_set_shard(rk.ctx, seeds_mode ? 0u : (uint32_t)r, seeds_mode ? 1u : world_n, mTileW, mTileH) != STHIP_OK) throw std::runtime_error(sthip_last_error(rk.ctx));
      sthip_outputs o{};
      o.device_ptrs = 1;
      o.gRayCount = (uint64_t*)b.counters;
      if (seeds_mode) {
        const uint32_t s0 = first_of((uint32_t)r), n = first_of((uint32_t)r + 1) - s0;
        o.radiance_layout = STHIP_LAYOUT_IMAGE;
        o.gRadiance = (float*)b.full;
        if (r == 0 && mGatherAOVs) {  // the G-buffer does not depend on the seed: rank 0's is the frame's
          o.gAlbedo = (float*)b.img_albedo;
          o.gVisibility = (VisibilityInfo*)b.img_visibility;
          o.gDepth = (DepthInfo*)b.img_depth;
          o.gPrevUVs = (float*)b.img_prev_uv;
        }
        if (n == 0) {  // more ranks than seeds: this one adds nothing
          check_hip(hipMemsetAsync(b.full, 0, pixels * 16, rk.stream), "hipMemsetAsync");
          check_hip(hipMemsetAsync(b.counters, 0, 16, rk.stream), "hipMemsetAsync");
        } else {
          if (sthip_render(rk.ctx, &fs.pc, sampling_flags(), fs.scene_flags, &fs.f, seed_begin + s0, n, &o) != STHIP_OK)
            throw std::runtime_error(std::string("sthip_render (rank ") + std::to_string(r) + "): " + sthip_last_error(rk.ctx));
          if (sthip_radiance_to_sums(rk.ctx, (float*)b.full, pixels, 0) != STHIP_OK) throw std::runtime_error(std::string("sthip_radiance_to_sums: ") + sthip_last_error(rk.ctx));
        }
      } else {
        o.radiance_layout = STHIP_LAYOUT_SHARD_TILES;
        o.gRadiance = (float*)b.radiance;
        if (mGatherAOVs) {
          o.gAlbedo = (float*)b.img_albedo;
          o.gVisibility = (VisibilityInfo*)b.img_visibility;
          o.gDepth = (DepthInfo*)b.img_depth;
          o.gPrevUVs = (float*)b.img_prev_uv;
        }
        if (sthip_render(rk.ctx, &fs.pc, sampling_flags(), fs.scene_flags, &fs.f, seed_begin, seed_count, &o) != STHIP_OK)
          throw std::runtime_error(std::string("sthip_render (rank ") + std::to_string(r) + "): " + sthip_last_error(rk.ctx));
        if (mGatherAOVs) {  // the G-buffer images -> this rank's tiles in slot order, like the radiance
          const void* img[4] = {b.img_albedo, b.img_visibility, b.img_depth, b.img_prev_uv};
          void* pk[4] = {b.albedo, b.visibility, b.depth, b.prev_uv};
          for (int a = 0; a < 4; a++)
            if (sthip_pack_tiles(rk.ctx, img[a], width, height, kEntryBytes[a + 1], pk[a]) != STHIP_OK) throw std::runtime_error(std::string("sthip_pack_tiles: ") + sthip_last_error(rk.ctx));
        }
      }
      check_hip(hipMemcpyAsync(rk.ray_count[k], b.counters, 16, hipMemcpyDeviceToHost, rk.stream), "hipMemcpyAsync");
      check_hip(hipEventRecord(rk.rendered[k], rk.stream), "hipEventRecord");
    });
    // ---- phase 2: the exchange, on the communication streams, behind the renders. A rank that fails in here takes every
    // communicator down with it (abort_comms), which ends the calls its peers are blocked in: nobody waits for ever ----
    try {
      mThreads->run(0, [&](size_t r) {
        try {
          Rank& rk = mRanks[r];
          Buffers& b = rk.buf[k];
          check_hip(hipSetDevice(mDevices[r]), "hipSetDevice");
          check_hip(hipStreamWaitEvent(rk.comm_stream, rk.rendered[k], 0), "hipStreamWaitEvent");
          CommUse use(*this, r);  // (throws if a peer has aborted the communicators already; an abort that comes later ends the calls below)
          if (seeds_mode) {
            // (into this frame's own buffer on rank 0 — the gather's, large enough: world x stride >= pixels — so that a frame in
            // flight never writes what the one before it is still being read from)
            check_nccl(ncclReduce(b.full, mGathered[k].radiance, 4 * pixels, ncclFloat, ncclSum, 0, use.comm, rk.comm_stream), "ncclReduce");
          } else {
            const void* src[5] = {b.radiance, b.albedo, b.visibility, b.depth, b.prev_uv};
            void* dst[5] = {mGathered[k].radiance, mGathered[k].albedo, mGathered[k].visibility, mGathered[k].depth, mGathered[k].prev_uv};
            const int parts = mGatherAOVs ? 5 : 1;
            check_nccl(ncclGroupStart(), "ncclGroupStart");
            for (int a = 0; a < parts; a++) {
              const size_t bytes = stride * kEntryBytes[a];
              check_nccl(ncclSend(src[a], bytes, ncclChar, 0, use.comm, rk.comm_stream), "ncclSend");
              if (r == 0)
                for (size_t q = 0; q < mRanks.size(); q++) check_nccl(ncclRecv((char*)dst[a] + q * bytes, bytes, ncclChar, (int)q, use.comm, rk.comm_stream), "ncclRecv");
            }
            check_nccl(ncclGroupEnd(), "ncclGroupEnd");
          }
          check_hip(hipEventRecord(rk.exchanged[k], rk.comm_stream), "hipEventRecord");
        } catch (...) {
          abort_comms();
          throw;
        }
      });
    } catch (...) {
      abort_comms();
      throw;
    }
    mInFlight[k].seeds_mode = seeds_mode;
    mInFlight[k].valid = true;
    mInFlight[k].aovs = mGatherAOVs;
    mInFlight[k].width = width;
    mInFlight[k].height = height;
    mInFlight[k].seed_count = seed_count;
    mInFlight[k].fs = fs;
    mInFlight[k].fs.f = sthip_frame_desc{};  // (its pointers lead into `fs`, which ends with this call; finish() needs the vectors only)
    mSubmitted++;
    note_submitted(fs, seed_count);
    if (!mPipelined)
      finish(k);
    else if (mInFlight[k ^ 1].valid)
      finish(k ^ 1);  // the frame before this one: its exchange, assembly and read-back run while this one renders
  }
  // completes the frames still in flight (pipelined mode): prev_result() is then the last frame submitted
  void flush() {
    for (int n = 0; n < 2; n++) {
      const int k = (int)((mSubmitted + n) & 1u);  // oldest first
      if (mInFlight[k].valid) finish(k);
    }
  }
  // the assembled frame on devices[0], valid until the next frame is finished: RGBA32F W x H, and the G-buffer
  const float* device_frame() const { return (const float*)mFrameDev.radiance; }
  const float* device_albedo() const { return (const float*)mFrameDev.albedo; }
  const VisibilityInfo* device_visibility() const { return (const VisibilityInfo*)mFrameDev.visibility; }
  const DepthInfo* device_depth() const { return (const DepthInfo*)mFrameDev.depth; }
  const float* device_prev_uv() const { return (const float*)mFrameDev.prev_uv; }

 private:
  static constexpr size_t kEntryBytes[5] = {16, 16, 8, 16, 8};  // radiance, albedo, VisibilityInfo, DepthInfo, prev-uv
  struct Buffers {  // one set per frame in flight, per rank: packed tiles (what travels) and the G-buffer images sthip_render writes
    void *radiance = nullptr, *albedo = nullptr, *visibility = nullptr, *depth = nullptr, *prev_uv = nullptr;
    void *img_albedo = nullptr, *img_visibility = nullptr, *img_depth = nullptr, *img_prev_uv = nullptr;
    void* full = nullptr;  // split_seeds: this rank's whole-frame radiance (sums over its seeds), what the reduce adds up
    void* counters = nullptr;
    std::vector<void*> all() const { return {radiance, albedo, visibility, depth, prev_uv, img_albedo, img_visibility, img_depth, img_prev_uv, full, counters}; }
  };
  struct Rank {
    sthip_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr, comm_stream = nullptr;
    hipEvent_t rendered[2] = {nullptr, nullptr}, exchanged[2] = {nullptr, nullptr};
    Buffers buf[2];
    uint64_t ray_count[2][2] = {{0, 0}, {0, 0}};
  };
  struct InFlight {
    bool valid = false, seeds_mode = false, aovs = true;
    uint32_t width = 0, height = 0, seed_count = 1;
    FrameSetup fs;
  };
  static void check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
  }
  static void check_nccl(ncclResult_t e, const char* what) {
    if (e != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(e));
  }
  void init_comms() {
    std::vector<ncclComm_t> comms(mDevices.size());
    check_nccl(ncclCommInitAll(comms.data(), (int)mDevices.size(), mDevices.data()), "ncclCommInitAll");
    for (size_t r = 0; r < mDevices.size(); r++) mRanks[r].comm = comms[r];
    mCommsAborted.store(false);
  }
  // Ends every collective in flight and every call a rank is blocked in; callable from any rank thread, once per failure
  // (the communicators are gone afterwards and made again by the next render()). A rank thread takes its communicator
  // through CommUse: under the rank's mutex it either sees the abort and throws, or is marked as inside its calls — so a
  // communicator is never aborted (freed) between a thread's look at it and that thread's next call on it.
  void abort_comms() {
    std::lock_guard<std::mutex> lk(mAbortMutex);
    if (mCommsAborted.exchange(true)) return;
    for (size_t r = 0; r < mRanks.size(); r++) {
      std::lock_guard<std::mutex> lr(mCommMutex[r]);
      if (mRanks[r].comm) {
        (void)ncclCommAbort(mRanks[r].comm);
        mRanks[r].comm = nullptr;
      }
    }
  }
  struct CommUse {
    MultiDeviceBDPT& d;
    size_t r;
    ncclComm_t comm;
    CommUse(MultiDeviceBDPT& driver, size_t rank) : d(driver), r(rank), comm(nullptr) {
      std::lock_guard<std::mutex> lk(d.mCommMutex[r]);
      if (d.mCommsAborted.load() || !d.mRanks[r].comm) throw std::runtime_error("the exchange was aborted (a peer failed in it)");
      comm = d.mRanks[r].comm;
    }
  };
  // the exchange of frame k has completed on rank 0 — or a communicator reports an asynchronous error (a peer died, a link
  // went down), in which case everything is aborted and the call throws instead of waiting for an event that never comes
  void wait_for_exchange(int k) {
    Rank& r0 = mRanks[0];
    for (;;) {
      const hipError_t q = hipEventQuery(r0.exchanged[k]);
      if (q == hipSuccess) return;
      if (q != hipErrorNotReady) check_hip(q, "hipEventQuery");
      for (size_t r = 0; r < mRanks.size(); r++) {
        ncclResult_t async = ncclSuccess;
        if (mRanks[r].comm && ncclCommGetAsyncError(mRanks[r].comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
          abort_comms();
          mInFlight[0].valid = mInFlight[1].valid = false;
          throw std::runtime_error(std::string("RCCL reports an asynchronous error on rank ") + std::to_string(r) + ": " + ncclGetErrorString(async));
        }
      }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  // frame k has been rendered and exchanged (or will have been when the events say so): assemble it on rank 0, read it back
  void finish(int k) {
    InFlight& f = mInFlight[k];
    f.valid = false;
    const uint32_t width = f.width, height = f.height;
    const size_t pixels = (size_t)width * height;
    const ShardLayout layout{width, height, (uint32_t)mRanks.size(), mTileW, mTileH};
    const size_t stride = layout.slot_count(0);
    Rank& r0 = mRanks[0];
    check_hip(hipSetDevice(mDevices[0]), "hipSetDevice");
    wait_for_exchange(k);
    // the assembly and the read-back run on rank 0's communication stream, behind the exchange: in pipelined mode rank 0's render
    // stream already holds the next frame, and nothing here should wait for it
    (void)sthip_set_stream(r0.ctx, r0.comm_stream);
    struct Restore {
      sthip_ctx* c;
      void* s;
      ~Restore() { (void)sthip_set_stream(c, s); }
    } restore{r0.ctx, r0.stream};
    const bool aovs = f.aovs;
    if (f.seeds_mode) {
      check_hip(hipMemcpyAsync(mFrameDev.radiance, mGathered[k].radiance, pixels * 16, hipMemcpyDeviceToDevice, r0.comm_stream), "hipMemcpyAsync");
      if (sthip_radiance_to_sums(r0.ctx, (float*)mFrameDev.radiance, pixels, 1) != STHIP_OK) throw std::runtime_error(sthip_last_error(r0.ctx));
      if (aovs) {  // rank 0's render wrote the G-buffer (its comm stream is behind that render: phase 2 made it wait)
        const Buffers& b0 = r0.buf[k];
        check_hip(hipMemcpyAsync(mFrameDev.albedo, b0.img_albedo, pixels * 16, hipMemcpyDeviceToDevice, r0.comm_stream), "hipMemcpyAsync");
        check_hip(hipMemcpyAsync(mFrameDev.visibility, b0.img_visibility, pixels * 8, hipMemcpyDeviceToDevice, r0.comm_stream), "hipMemcpyAsync");
        check_hip(hipMemcpyAsync(mFrameDev.depth, b0.img_depth, pixels * 16, hipMemcpyDeviceToDevice, r0.comm_stream), "hipMemcpyAsync");
        check_hip(hipMemcpyAsync(mFrameDev.prev_uv, b0.img_prev_uv, pixels * 8, hipMemcpyDeviceToDevice, r0.comm_stream), "hipMemcpyAsync");
      }
    } else {
      const void* src[5] = {mGathered[k].radiance, mGathered[k].albedo, mGathered[k].visibility, mGathered[k].depth, mGathered[k].prev_uv};
      void* dst[5] = {mFrameDev.radiance, mFrameDev.albedo, mFrameDev.visibility, mFrameDev.depth, mFrameDev.prev_uv};
      const int parts = aovs ? 5 : 1;
      for (int a = 0; a < parts; a++)
        if (sthip_assemble_tiles_bytes(r0.ctx, src[a], stride, (uint32_t)mRanks.size(), mTileW, mTileH, width, height, (uint32_t)kEntryBytes[a], dst[a]) != STHIP_OK)
          throw std::runtime_error(sthip_last_error(r0.ctx));
    }
    Frame fr;
    fr.width = width;
    fr.height = height;
    if (mReadBack) {
      fr.mRadiance.assign(4 * pixels, 0.f);
      check_hip(hipMemcpyAsync(fr.mRadiance.data(), mFrameDev.radiance, pixels * 16, hipMemcpyDeviceToHost, r0.comm_stream), "hipMemcpyAsync");
      if (aovs) {
        fr.mAlbedo.assign(4 * pixels, 0.f);
        fr.mVisibility.assign(pixels, VisibilityInfo{});
        fr.mDepth.assign(pixels, DepthInfo{});
        fr.mPrevUVs.assign(2 * pixels, 0.f);
        check_hip(hipMemcpyAsync(fr.mAlbedo.data(), mFrameDev.albedo, pixels * 16, hipMemcpyDeviceToHost, r0.comm_stream), "hipMemcpyAsync");
        check_hip(hipMemcpyAsync(fr.mVisibility.data(), mFrameDev.visibility, pixels * 8, hipMemcpyDeviceToHost, r0.comm_stream), "hipMemcpyAsync");
        check_hip(hipMemcpyAsync(fr.mDepth.data(), mFrameDev.depth, pixels * 16, hipMemcpyDeviceToHost, r0.comm_stream), "hipMemcpyAsync");
        check_hip(hipMemcpyAsync(fr.mPrevUVs.data(), mFrameDev.prev_uv, pixels * 8, hipMemcpyDeviceToHost, r0.comm_stream), "hipMemcpyAsync");
      }
    }
    check_hip(hipStreamSynchronize(r0.comm_stream), "hipStreamSynchronize");
    for (size_t r = 0; r < mRanks.size(); r++) {  // the ray counts: every rank's render stream has passed its copy when its `rendered` event has
      check_hip(hipSetDevice(mDevices[r]), "hipSetDevice");
      check_hip(hipEventSynchronize(mRanks[r].rendered[k]), "hipEventSynchronize");
      fr.mRayCount[0] += mRanks[r].ray_count[k][0];
      fr.mRayCount[1] += mRanks[r].ray_count[k][1];
    }
    check_hip(hipSetDevice(mDevices[0]), "hipSetDevice");
    finish_frame(std::move(fr), f.fs, f.seed_count, true);  // (frame number and previous views moved on when the frame was submitted)
  }
  void ensure_buffers(size_t stride, size_t pixels) {
    if (stride > mStride || pixels > mPixels) {
      flush();
      const size_t s = std::max(stride, mStride), px = std::max(pixels, mPixels);
      auto renew = [&](void*& p, size_t bytes) {
        if (p) (void)hipFree(p);
        p = nullptr;
        check_hip(hipMalloc(&p, std::max<size_t>(bytes, 16)), "hipMalloc");
        check_hip(hipMemset(p, 0, std::max<size_t>(bytes, 16)), "hipMemset");
      };
      for (size_t r = 0; r < mRanks.size(); r++) {
        check_hip(hipSetDevice(mDevices[r]), "hipSetDevice");
        for (int k = 0; k < 2; k++) {
          Buffers& b = mRanks[r].buf[k];
          renew(b.radiance, s * 16);
          renew(b.albedo, s * 16);
          renew(b.visibility, s * 8);
          renew(b.depth, s * 16);
          renew(b.prev_uv, s * 8);
          renew(b.img_albedo, px * 16);
          renew(b.img_visibility, px * 8);
          renew(b.img_depth, px * 16);
          renew(b.img_prev_uv, px * 8);
          renew(b.full, px * 16);
          renew(b.counters, 16);
        }
      }
      check_hip(hipSetDevice(mDevices[0]), "hipSetDevice");
      const size_t w = mRanks.size();
      for (int k = 0; k < 2; k++) {
        renew(mGathered[k].radiance, w * s * 16);
        renew(mGathered[k].albedo, w * s * 16);
        renew(mGathered[k].visibility, w * s * 8);
        renew(mGathered[k].depth, w * s * 16);
        renew(mGathered[k].prev_uv, w * s * 8);
      }
      renew(mFrameDev.radiance, px * 16);
      renew(mFrameDev.albedo, px * 16);
      renew(mFrameDev.visibility, px * 8);
      renew(mFrameDev.depth, px * 16);
      renew(mFrameDev.prev_uv, px * 8);
      mStride = s;
      mPixels = px;
    }
  }
  std::vector<int> mDevices;
  std::vector<Rank> mRanks;
  std::unique_ptr<RankThreads> mThreads;
  uint32_t mTileW, mTileH;
  Buffers mGathered[2];  // on devices[0]: rank r's tiles at r * stride entries (only the packed members are used)
  Buffers mFrameDev;     // on devices[0]: the assembled W x H images (radiance, albedo, visibility, depth, prev_uv)
  InFlight mInFlight[2];
  uint64_t mSubmitted = 0;
  bool mGatherAOVs = true, mPipelined = false, mSplitSeeds = false, mReadBack = true;
  std::atomic<bool> mCommsAborted{false};
  std::mutex mAbortMutex;
  std::vector<std::mutex> mCommMutex;  // per rank: its communicator pointer (CommUse / abort_comms)
  size_t mStride = 0, mPixels = 0;
};

}  // namespace stm
