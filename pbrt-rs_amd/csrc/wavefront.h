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

int wavefront_render(PbrtHipScene* s, const PbrtCamera& camera, const PbrtRenderParams& rp, float* d_film,
                     PbrtRenderStats* stats);
