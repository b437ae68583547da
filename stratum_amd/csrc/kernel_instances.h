// kernel_instances.h — which instantiations of the big kernel templates (kernels.h) live in which translation unit. api.hip only
// launches them: it sees `extern template` declarations (STHIP_DECLARE_KERNEL_INSTANCES) and the definitions are compiled, in
// parallel, in shade_*.hip and trace_kernels.hip (__graft_entry__.build_product): one hipcc process per file instead of four
// minutes of one. A launch of an instantiation that is not listed here still works — it is then compiled into api.hip itself.
#pragma once
#include "kernels.h"

// X(TEXTURED, EXT, LT, MEDIA, PROBE, DEBUG)
#define STHIP_SHADE_PLAIN(X) \
  X(false, false, false, 0, false, false) X(true, false, false, 0, false, false) X(false, true, false, 0, false, false) X(true, true, false, 0, false, false) \
  X(false, true, false, 0, true, false) X(true, true, false, 0, true, false) X(true, true, false, 0, false, true)
#define STHIP_SHADE_LT(X) \
  X(false, true, true, 0, false, false) X(true, true, true, 0, false, false) X(false, true, true, 0, true, false) X(true, true, true, 0, true, false) X(true, true, true, 0, false, true)
#define STHIP_SHADE_MEDIA(X) \
  X(false, true, false, 1, false, false) X(true, true, false, 1, false, false) X(true, true, false, 1, false, true) X(false, true, false, 2, false, false) X(true, true, false, 2, false, false) \
  X(true, true, false, 2, false, true)
#define STHIP_SHADE_MEDIA_LT(X) X(false, true, true, 1, false, false) X(true, true, true, 1, false, false) X(true, true, true, 1, false, true)
#define STHIP_SHADE_MEDIA_LT2(X) X(false, true, true, 2, false, false) X(true, true, true, 2, false, false) X(true, true, true, 2, false, true)
// Y(COUNT, ALPHA, BOUNDED, TOP, WIDE)
#define STHIP_TRACE_ROWS(Y, TOP, WIDE) \
  Y(false, false, false, TOP, WIDE) Y(true, false, false, TOP, WIDE) Y(false, true, false, TOP, WIDE) Y(true, true, false, TOP, WIDE) Y(false, false, true, TOP, WIDE) Y(true, false, true, TOP, WIDE) \
  Y(false, true, true, TOP, WIDE) Y(true, true, true, TOP, WIDE)
#define STHIP_TRACE_ALL(Y) STHIP_TRACE_ROWS(Y, false, 0) STHIP_TRACE_ROWS(Y, true, 0) STHIP_TRACE_ROWS(Y, false, 1) STHIP_TRACE_ROWS(Y, false, 2)
// Z(TEXTURED, EXT, MEDIA)
#define STHIP_SHADE_LIGHT(Z) Z(false, true, false) Z(true, true, false) Z(false, true, true) Z(true, true, true)

#define STHIP_SHADE_EXTERN(T, E, L, M, P, D) extern template __global__ void k_shade<T, E, L, M, P, D>(FrameParams, uint32_t);
#define STHIP_SHADE_DEFINE(T, E, L, M, P, D) template __global__ void k_shade<T, E, L, M, P, D>(FrameParams, uint32_t);
#define STHIP_TRACE_EXTERN(C, A, B, T, W) extern template __global__ void k_trace<C, A, B, T, W>(FrameParams, uint32_t, uint32_t);
#define STHIP_TRACE_DEFINE(C, A, B, T, W) template __global__ void k_trace<C, A, B, T, W>(FrameParams, uint32_t, uint32_t);
#define STHIP_LIGHT_EXTERN(T, E, M) extern template __global__ void k_shade_light<T, E, M>(FrameParams, uint32_t);
#define STHIP_LIGHT_DEFINE(T, E, M) template __global__ void k_shade_light<T, E, M>(FrameParams, uint32_t);

#define STHIP_DECLARE_KERNEL_INSTANCES \
  STHIP_SHADE_PLAIN(STHIP_SHADE_EXTERN) STHIP_SHADE_LT(STHIP_SHADE_EXTERN) STHIP_SHADE_MEDIA(STHIP_SHADE_EXTERN) STHIP_SHADE_MEDIA_LT(STHIP_SHADE_EXTERN) STHIP_SHADE_MEDIA_LT2(STHIP_SHADE_EXTERN) STHIP_TRACE_ALL(STHIP_TRACE_EXTERN) \
      STHIP_SHADE_LIGHT(STHIP_LIGHT_EXTERN)
