// wavefront.h — wavefront path tracer: SamplerIntegrator::render (src/core/integrator.rs:399-480)
// + PathIntegrator::li (src/integrators/path.rs:65-213) + uniform_sample_one_light / estimate_direct
// (src/core/integrator.rs:92-266) as a sequence of kernels over SoA path state in HBM.
//
// One pass handles `spp_per_pass` samples of every pixel of this GPU's tiles concurrently
// (N = pixels x samples paths). Per bounce: k_trace traces every pending ray of every path
// (continuation and MIS rays = Scene::intersect, shadow rays = Scene::intersect_p) in ONE launch;
// k_shade then, per path, (1) resolves the previous bounce's direct-lighting estimate from the
// shadow / MIS results, (2) processes the continuation hit: emission, SurfaceInteraction
// (src/shapes/triangle.rs:193-250), BSDF, light sampling, BSDF sampling, Russian roulette, and
// appends the next rays with wave-aggregated queue appends. Random numbers are drawn in the
// reference's order from the path's own PCG32 stream, so every decision matches the CPU path.
#pragma once
#include <hip/hip_runtime.h>

// the parts, in dependency order
#include "wf_state.h"
#include "wf_sampler.h"
#include "wf_generate_trace.h"
#include "wf_surface.h"
#include "wf_lights.h"
#include "wf_path.h"
#include "wf_direct.h"
#include "wf_film.h"

// The same loop driven from outside the render form:
//   li:     a batch of Integrator::li calls on the caller's rays (pbrt_hip_li): rays / keys in, radiance out, no film;
//   export: only the camera-ray stage of a render (pbrt_hip_camera_rays): what k_generate made, copied out.
struct LiBatch {
    const PbrtRay* d_rays = nullptr;   // li: n caller rays (device)
    const uint64_t* d_keys = nullptr;  // li: RNG::set_sequence argument per ray (device)
    int64_t n = 0;
    int skip = 0;                      // values already drawn from each stream before li
    float* d_rgb = nullptr;            // li: n x 3 (device)
    PbrtRay* out_rays = nullptr;       // export (device): per path of the single pass
    uint64_t* out_keys = nullptr;
    float* out_pfilm = nullptr;
    int32_t* out_pixel_sample = nullptr;
};
int wavefront_render(PbrtHipScene* s, const PbrtCamera& camera, const PbrtRenderParams& rp, float* d_film,
                     PbrtRenderStats* stats, const LiBatch* li = nullptr);
