// fused_kernel.h — placeholder until the streaming kernel lands.
#pragma once
#include "dsp_device.h"
namespace rtlfm { namespace fused {
struct Workspace { void release() {} };
inline bool supported(const rtlfm_cfg &, int) { return false; }
inline int launch(Workspace &, const rtlfm_cfg &, int, const uint8_t *, size_t, int, int16_t *, size_t,
                  const rtlfm_stream_state *, rtlfm_stream_state *, const int32_t *, hipStream_t) { return -ENOTSUP; }
}}
