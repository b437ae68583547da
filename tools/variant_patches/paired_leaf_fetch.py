p='traverse.h'
s=open(p).read()
old=s[s.index("    for (uint32_t i = 0; i < count; i++) {\n      const float4* tv = reinterpret_cast<const float4*>(tbase + (size_t)((first + i) * 48u));"):s.index("    // pop — or, for an occlusion lane that found its hit, the end of the ray")]
new='''    for (uint32_t i = 0; i < count; i += 2) {
      // two triangles per trip: both fetches are in flight before the first test (a leaf of the SAH builder has at most two)
      const bool second = i + 1 < count;
      const float4* tv = reinterpret_cast<const float4*>(tbase + (size_t)((first + i) * 48u));
      const float4* tw = reinterpret_cast<const float4*>(tbase + (size_t)((first + i + (second ? 1u : 0u)) * 48u));
      const float4 v0 = tv[0], v1 = tv[1], v2 = tv[2];
      const float4 w0 = tw[0], w1 = tw[1], w2 = tw[2];
      if (COUNT) {
        cnt.tris += second ? 2 : 1;
        if (first_active_lane()) cnt.tri_slots += 64;
      }
      const bool any_lane = is_any();
      {
        float t, b1, b2;
        bool candidate = tri_test(sp, xyz(v0), xyz(v1), xyz(v2), tmin, tmax, t, b1, b2);
        if (ALPHA && candidate) candidate = alpha_pass(bvh, first + i, __float_as_uint(v0.w) | id_bits, b1, b2);
        occluded |= candidate & any_lane;
        const uint32_t ip = __float_as_uint(v0.w) | id_bits;
        const bool closer = candidate & !any_lane & ((t < hit.t) | ((t == hit.t) & (hit.ip != 0xFFFFFFFFu) & (hit_key(ip) < hit_key(hit.ip))));
        hit.t = closer ? t : hit.t;
        hit.b1 = closer ? b1 : hit.b1;
        hit.b2 = closer ? b2 : hit.b2;
        hit.ip = closer ? ip : hit.ip;
      }
      __builtin_amdgcn_sched_barrier(0);  // the second test behind the first: interleaved they need 21 registers more
      {
        float t, b1, b2;
        bool candidate = tri_test(sp, xyz(w0), xyz(w1), xyz(w2), tmin, tmax, t, b1, b2) & second;
        if (ALPHA && candidate) candidate = alpha_pass(bvh, first + i + 1, __float_as_uint(w0.w) | id_bits, b1, b2);
        occluded |= candidate & any_lane;
        const uint32_t ip = __float_as_uint(w0.w) | id_bits;
        const bool closer = candidate & !any_lane & ((t < hit.t) | ((t == hit.t) & (hit.ip != 0xFFFFFFFFu) & (hit_key(ip) < hit_key(hit.ip))));
        hit.t = closer ? t : hit.t;
        hit.b1 = closer ? b1 : hit.b1;
        hit.b2 = closer ? b2 : hit.b2;
        hit.ip = closer ? ip : hit.ip;
      }
    }
'''
s=s.replace(old,new)
# alpha helper
helper='''  // gAlphaTest: the candidate must pass the mask of its instance's material (instances that share a mesh may
  // have different materials, so the mask comes from the instance, the uvs from the leaf triangle)
  DEV bool alpha_pass(const DeviceBvh& bvh, uint32_t tri, uint32_t ip, float b1, float b2) const {
    const uint32_t mask = bvh.alpha_test ? bvh.inst_alpha[ip & 0xFFFFu] : BVH_NO_ALPHA;
    if (mask == BVH_NO_ALPHA) return true;
    const float2* q = bvh.tri_uv + (size_t)tri * 3u;
    const float2 u0 = q[0], u1 = q[1], u2 = q[2];
    const float u = u0.x + (u1.x - u0.x) * b1 + (u2.x - u0.x) * b2;  // shading_data.hlsli:2-6
    float v = u0.y + (u1.y - u0.y) * b1 + (u2.y - u0.y) * b2;
    if (bvh.flip_uvs) v = 1 - v;
    return sample_image1(bvh, mask, u, v) >= 0.75f;
  }

'''
anchor="  // ref has the leaf bit: a sentinel, an instance, or up to 4 triangles\n"
assert anchor in s
s=s.replace(anchor,helper+anchor)
open(p,'w').write(s)
