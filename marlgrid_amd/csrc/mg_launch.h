// mg_launch.h — host-side launchers implemented by the individual kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "marlgrid_hip.h"

namespace mg {
hipError_t launch_mt_seed(int B, const uint32_t* keys, const int32_t* key_len, uint32_t* mt, int32_t* mt_pos,
                          uint32_t* mt_head, hipStream_t s);
hipError_t launch_reset(const MgConfig& cfg, const MgState& st, const MgGenProgram& prog, const uint8_t* mask,
                        hipStream_t s);
hipError_t launch_step(const MgConfig& cfg, const MgState& st, const void* actions, int action_bytes,
                       float* rewards, const MgGenProgram* auto_reset, hipStream_t s);
struct FusedStep;
// which instantiation of mg::render_kernel<VS, TS, WPB, V, RM> a configuration gets (0 = the value is read from the
// config at run time) and the LDS bytes of one of its workgroups: filled in INSTEAD of launching when handed to launch_render
struct RenderPick { int vs, ts, wpb, v, rm, lds; };
hipError_t launch_render(const MgConfig& cfg, const MgState& st, uint8_t* obs, uint8_t* view_cells,
                         uint8_t* view_agent, uint8_t* vis_mask, hipStream_t s, const FusedStep* fused_step = nullptr,
                         RenderPick* pick = nullptr);
int render_min_lds_bytes(const MgConfig& cfg);
bool render_can_encode(const MgConfig& cfg);      // mg_step_render_encode: this configuration's step launch can write the encoding too
hipError_t launch_encode(const MgConfig& cfg, const MgState& st, const uint8_t* vis_mask, uint8_t* out,
                         hipStream_t s);
hipError_t launch_put_obj(const MgConfig& cfg, const MgState& st, int obj, int x, int y, const uint8_t* mask,
                          hipStream_t s);
hipError_t launch_place(const MgConfig& cfg, const MgState& st, int what, int x0, int y0, int x1, int y1, int max_tries,
                        const int32_t* fixed_pos, const uint8_t* mask, const uint8_t* reject, int32_t* out_pos,
                        uint8_t* out_ok, hipStream_t s);
hipError_t launch_frame(const MgConfig& cfg, const MgState& st, const int32_t* env_ids, int K, const uint8_t* atlas,
                        int ts, int highlight, uint32_t amax, uint8_t* out, hipStream_t s);
}  // namespace mg
