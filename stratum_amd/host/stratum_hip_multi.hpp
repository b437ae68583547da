// stratum_hip_multi.hpp — the multi-GPU driver of the C++ host side: one process, one sthip context per GPU, one host
// thread per context, the frame cut into pixel tiles (sthip_set_shard), and ONE exchange per render call: an RCCL gather
// of the ranks' packed tiles to the first device over xGMI, scattered into the image there (sthip_assemble_tiles).
//
// The reference has nothing to compare with: it creates one vk::Device with one queue per family and renders on it
// (src/Core/Device.cpp:125-131). This is the component BASELINE.json's north star adds ("frames shard by pixel-tile
// across the 8 GPUs of one node with an RCCL reduce of the accumulation buffer over xGMI") on the host language the
// north star keeps (C++). The path shards without any data-path communication: a pixel's RNG key is
// (x, y, seed, counter) (rng.hlsli:35-47) and a pixel owns its outputs, so the ranks only meet to assemble the frame.
//
// stm::MultiDeviceBDPT is a stm::BDPT (same component, same update()/render() the Application drives, BDPT.hpp:23-24):
// its base context renders rank 0, the extra contexts the other ranks. Needs <rccl/rccl.h> and the HIP runtime
// (link -lrccl -lamdhip64); stratum_hip.hpp itself stays free of both.
#pragma once

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

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
  // what sthip_assemble_tiles does on the device, on the host: packed[r] = rank r's buffer (slot_count(r) float4 entries)
  void assemble(const std::vector<const float*>& packed, float* frame) const {
    std::memset(frame, 0, (size_t)width * height * 16);
    for (uint32_t r = 0; r < world; r++)
      for (uint32_t s = 0, n = slot_count(r); s < n; s++) {
        uint32_t x, y;
        if (slot_pixel(r, s, x, y)) std::memcpy(frame + 4 * ((size_t)y * width + x), packed[r] + 4 * (size_t)s, 16);
      }
  }
};

class MultiDeviceBDPT : public BDPT {
 public:
  // devices[0] is rank 0 (the base class's context): it receives the tiles and holds the assembled frame
  explicit MultiDeviceBDPT(Node& node, const std::vector<int>& devices, uint32_t tile_w = 64, uint32_t tile_h = 32)
      : BDPT(node, devices.empty() ? 0 : devices[0]), mDevices(devices), mTileW(tile_w), mTileH(tile_h) {
    if (devices.empty()) throw std::invalid_argument("MultiDeviceBDPT: no devices");
    mRanks.resize(devices.size());
    mRanks[0].ctx = mCtx;
    for (size_t r = 1; r < devices.size(); r++)
      if (sthip_create(devices[r], &mRanks[r].ctx) != STHIP_OK) throw std::runtime_error(std::string("sthip_create: ") + sthip_last_error(nullptr));
    std::vector<ncclComm_t> comms(devices.size());
    check_nccl(ncclCommInitAll(comms.data(), (int)devices.size(), devices.data()), "ncclCommInitAll");
    for (size_t r = 0; r < devices.size(); r++) {
      mRanks[r].comm = comms[r];
      check_hip(hipSetDevice(devices[r]), "hipSetDevice");
      check_hip(hipStreamCreateWithFlags(&mRanks[r].stream, hipStreamNonBlocking), "hipStreamCreate");
      (void)sthip_set_stream(mRanks[r].ctx, mRanks[r].stream);
    }
  }
  ~MultiDeviceBDPT() override {
    for (size_t r = 0; r < mRanks.size(); r++) {
      (void)hipSetDevice(mDevices[r]);
      (void)hipStreamSynchronize(mRanks[r].stream);
      if (mRanks[r].comm) (void)ncclCommDestroy(mRanks[r].comm);
      if (mRanks[r].packed) (void)hipFree(mRanks[r].packed);
      if (mRanks[r].counters) (void)hipFree(mRanks[r].counters);
      if (r > 0 && mRanks[r].ctx) sthip_destroy(mRanks[r].ctx);  // rank 0's context is the base class's
      if (mRanks[r].stream) (void)hipStreamDestroy(mRanks[r].stream);
    }
    (void)hipSetDevice(mDevices[0]);
    if (mGathered) (void)hipFree(mGathered);
    if (mFrameDev) (void)hipFree(mFrameDev);
  }
  size_t world() const { return mRanks.size(); }

  // the scene goes to every GPU (replicated: a 1M-triangle scene is < 200 MB of 288 GB), in parallel
  void update(CommandBuffer& cb, float dt) override {
    auto scene = mNode.find_in_ancestor<Scene>();
    if (!scene) scene = mNode.root().find_in_descendants<Scene>();
    if (!scene || !scene->data() || scene->data().get() == mBound) return;
    BDPT::update(cb, dt);  // rank 0, and the bookkeeping (mBound, light count, environment)
    (void)sthip_set_stream(mCtx, mRanks[0].stream);
    const sthip_scene_desc d = scene->data()->desc();
    const bool transforms_only = last_update_was_transforms_only();
    for_each_rank(1, [&](size_t r) {
      int rc = STHIP_ERR_UNSUPPORTED;
      if (transforms_only) rc = sthip_scene_update_transforms(mRanks[r].ctx, d.gInstanceTransforms, d.gInstanceInverseTransforms, d.gInstanceMotionTransforms, d.instance_count);
      if (rc != STHIP_OK && sthip_scene_upload(mRanks[r].ctx, &d) != STHIP_OK) throw std::runtime_error(std::string("sthip_scene_upload (rank ") + std::to_string(r) + "): " + sthip_last_error(mRanks[r].ctx));
    });
  }

  // One frame over all GPUs: every rank renders its tiles for all the seeds of the call (packed, 1 / world of the frame),
  // the tiles are gathered on rank 0 (one message per rank straight over its xGMI link: ncclSend / ncclRecv in a group,
  // no ring) and scattered into the image there. mPrevFrame holds the radiance and the ray counts; the G-buffer AOVs of a
  // sharded frame stay with the rank that owns the pixel and are not exchanged (render on one device when they are needed).
  void render(CommandBuffer&, uint32_t width, uint32_t height, const std::vector<std::pair<ViewData, TransformData>>& views, uint32_t seed_count = 1) override {
    FrameSetup fs;
    prepare_frame(width, height, views, fs);
    const ShardLayout layout{width, height, (uint32_t)mRanks.size(), mTileW, mTileH};
    const size_t stride = layout.slot_count(0);  // rank 0 owns the most tiles: equal-size messages
    ensure_buffers(stride, (size_t)width * height);
    const uint32_t seed_begin = frame_number();
    for_each_rank(0, [&](size_t r) {
      Rank& rk = mRanks[r];
      check_hip(hipSetDevice(mDevices[r]), "hipSetDevice");
      if (sthip_set_shard(rk.ctx, (uint32_t)r, (uint32_t)mRanks.size(), mTileW, mTileH) != STHIP_OK) throw std::runtime_error(sthip_last_error(rk.ctx));
      sthip_outputs o{};
      o.device_ptrs = 1;
      o.radiance_layout = STHIP_LAYOUT_SHARD_TILES;
      o.gRadiance = rk.packed;
      o.gRayCount = rk.counters;
      if (sthip_render(rk.ctx, &fs.pc, sampling_flags(), fs.scene_flags, &fs.f, seed_begin, seed_count, &o) != STHIP_OK)
        throw std::runtime_error(std::string("sthip_render (rank ") + std::to_string(r) + "): " + sthip_last_error(rk.ctx));
      // the exchange: behind the render on the rank's stream, so nothing on the host waits in between
      check_nccl(ncclGroupStart(), "ncclGroupStart");
      check_nccl(ncclSend(rk.packed, stride * 4, ncclFloat, 0, rk.comm, rk.stream), "ncclSend");
      if (r == 0)
        for (size_t q = 0; q < mRanks.size(); q++) check_nccl(ncclRecv(mGathered + q * stride * 4, stride * 4, ncclFloat, (int)q, rk.comm, rk.stream), "ncclRecv");
      check_nccl(ncclGroupEnd(), "ncclGroupEnd");
      check_hip(hipMemcpyAsync(rk.ray_count, rk.counters, 16, hipMemcpyDeviceToHost, rk.stream), "hipMemcpyAsync");
      if (r == 0) {
        if (sthip_assemble_tiles(rk.ctx, mGathered, stride, (uint32_t)mRanks.size(), mTileW, mTileH, width, height, mFrameDev) != STHIP_OK) throw std::runtime_error(sthip_last_error(rk.ctx));
      }
      check_hip(hipStreamSynchronize(rk.stream), "hipStreamSynchronize");
    });
    Frame fr;
    fr.width = width;
    fr.height = height;
    fr.mRadiance.assign(4 * (size_t)width * height, 0.f);
    check_hip(hipSetDevice(mDevices[0]), "hipSetDevice");
    check_hip(hipMemcpy(fr.mRadiance.data(), mFrameDev, fr.mRadiance.size() * 4, hipMemcpyDeviceToHost), "hipMemcpy");
    for (const Rank& rk : mRanks) {
      fr.mRayCount[0] += rk.ray_count[0];
      fr.mRayCount[1] += rk.ray_count[1];
    }
    finish_frame(std::move(fr), fs, seed_count);
  }
  const float* device_frame() const { return mFrameDev; }  // RGBA32F W x H on devices[0], valid until the next render

 private:
  struct Rank {
    sthip_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    float* packed = nullptr;  // this rank's tiles in slot order
    uint64_t* counters = nullptr;
    uint64_t ray_count[2] = {0, 0};
  };
  static void check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
  }
  static void check_nccl(ncclResult_t e, const char* what) {
    if (e != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(e));
  }
  // fn(rank) on one host thread per rank (a context is single-threaded, different contexts may run concurrently: sthip.h)
  template <typename F>
  void for_each_rank(size_t first, F&& fn) {
    std::vector<std::thread> pool;
    std::vector<std::string> errors(mRanks.size());
    for (size_t r = first; r < mRanks.size(); r++)
      pool.emplace_back([&, r]() {
        try {
          fn(r);
        } catch (const std::exception& e) {
          errors[r] = e.what();
        }
      });
    for (auto& t : pool) t.join();
    for (const std::string& e : errors)
      if (!e.empty()) throw std::runtime_error(e);
  }
  void ensure_buffers(size_t stride, size_t pixels) {
    if (stride > mStride) {
      for (size_t r = 0; r < mRanks.size(); r++) {
        check_hip(hipSetDevice(mDevices[r]), "hipSetDevice");
        if (mRanks[r].packed) (void)hipFree(mRanks[r].packed);
        check_hip(hipMalloc((void**)&mRanks[r].packed, stride * 16), "hipMalloc");
        if (!mRanks[r].counters) check_hip(hipMalloc((void**)&mRanks[r].counters, 16), "hipMalloc");
      }
      check_hip(hipSetDevice(mDevices[0]), "hipSetDevice");
      if (mGathered) (void)hipFree(mGathered);
      check_hip(hipMalloc((void**)&mGathered, mRanks.size() * stride * 16), "hipMalloc");
      mStride = stride;
    }
    if (pixels > mPixels) {
      check_hip(hipSetDevice(mDevices[0]), "hipSetDevice");
      if (mFrameDev) (void)hipFree(mFrameDev);
      check_hip(hipMalloc((void**)&mFrameDev, pixels * 16), "hipMalloc");
      mPixels = pixels;
    }
  }
  std::vector<int> mDevices;
  std::vector<Rank> mRanks;
  uint32_t mTileW, mTileH;
  float* mGathered = nullptr;
  float* mFrameDev = nullptr;
  size_t mStride = 0, mPixels = 0;
};

}  // namespace stm
