// Device BODY of the weight-image producer (see weight_images.hip for what the images are), shared by the stand-alone launch
// (grl_weight_images) and the step's merged head launch (node_ops.hip: grl_step_head).
#pragma once
#include "grl_wimg.h"

namespace {

constexpr int WIMG_MAX_JOBS = 24;
constexpr int WIMG_PARTS = 8;   // parts of one image, built by different workgroups
struct WimgJobs {
  int n;
  int kind[WIMG_MAX_JOBS];
  const float* src[WIMG_MAX_JOBS][6];   // kind 0 / 1: W1 b1 W2 b2 Wk grid;  kind 2: W3 b3 W4 b4 gamma beta;  kind 3: W3 W4
  void* out[WIMG_MAX_JOBS];
};

// workgroup `part` (of WIMG_PARTS) of image job `j`: 256 threads
GRL_DEVINL void weight_images_body(const WimgJobs& jobs, int j, int part) {
  const float* const* a = jobs.src[j];
  switch (jobs.kind[j]) {
    case WIMG_EDGE16: {
      Edge16Image& im = *reinterpret_cast<Edge16Image*>(jobs.out[j]);
      if (part == 0) {
        stage16<14, 32, 256>(im.w.W1h, im.w.W1l, a[0], WI_LD1);
        stage_chain16_small<256>(im.w, a[1], a[3], a[5]);
      } else if (part == 1) stage16<64, 64, 256>(im.w.W2h, im.w.W2l, a[2], WI_LD2);
      else if (part == 2) stage16<64, 64, 256>(im.w.Wkh, im.w.Wkl, a[4], WI_LD2);
      else if (part == 3) stage16<64, 64, 256, true>(im.WkTh, im.WkTl, a[4], WI_LD2);
      else if (part == 4) stage16<64, 64, 256, true>(im.W2Th, im.W2Tl, a[2], WI_LD2);
      break;
    }
    case WIMG_EDGE32: {
      ChainW& im = *reinterpret_cast<ChainW*>(jobs.out[j]);
      if (part == 0) {
        stage_split<64, 16, 14, 256>(im.W1h, im.W1l, a[0], WI_LDB1);
        for (int i = threadIdx.x; i < 64; i += 256) {
          im.b1s[i] = a[1][i];
          im.b2s[i] = a[3][i];
          im.grid_s[i] = i < 48 ? a[5][i] : 0.f;
        }
      } else if (part == 1) stage_split<64, 64, 64, 256>(im.W2h, im.W2l, a[2], WI_LDB);
      else if (part == 2) stage_split<64, 64, 64, 256>(im.Wkh, im.Wkl, a[4], WI_LDB);
      break;
    }
    case WIMG_MLP_FWD: {
      MlpSmemBf& im = *reinterpret_cast<MlpSmemBf*>(jobs.out[j]);
      if (part < 4) {   // W3 [256,64]: 64 rows per part
        stage_split<64, 64, 64, 256>(im.W3h + part * 64 * WI_LB3, im.W3l + part * 64 * WI_LB3, a[0] + part * 64 * 64, WI_LB3);
      } else if (part < 8) {   // W4 [64,256]: 16 rows per part
        const int p4 = part - 4;
        stage_split<16, 256, 256, 256>(im.W4h + p4 * 16 * WI_LB4, im.W4l + p4 * 16 * WI_LB4, a[2] + p4 * 16 * 256, WI_LB4);
        if (p4 == 0) {
          for (int i = threadIdx.x; i < 256; i += 256) im.b3s[i] = a[1][i];
          for (int i = threadIdx.x; i < 64; i += 256) { im.b4s[i] = a[3][i]; im.gam[i] = a[4][i]; im.bet[i] = a[5][i]; }
        }
      }
      break;
    }
    case WIMG_MLP_BWD16: {
      Mlp16Image& im = *reinterpret_cast<Mlp16Image*>(jobs.out[j]);
      if (part < 4) mlp16_fragments(im, a[0], a[1], part /*wave*/, threadIdx.x >> 6 /*n-tile*/, threadIdx.x & 63);
      break;
    }
    default: break;
  }
}

// kinds [n], srcs [n][6], outs [n] (HOST arrays: include/grl_hip.h grl_weight_images) -> jobs; 0 or a negative status
inline int wimg_fill(WimgJobs& jobs, int n, const int* kinds, const float* const* srcs, void* const* outs) {
  if (n > WIMG_MAX_JOBS) return -2;
  jobs.n = n;
  for (int j = 0; j < n; ++j) {
    if (kinds[j] < 0 || kinds[j] >= WIMG_KINDS || !outs[j]) return -3;
    jobs.kind[j] = kinds[j];
    jobs.out[j] = outs[j];
    const int need = kinds[j] == WIMG_MLP_BWD16 ? 2 : 6;
    for (int k = 0; k < 6; ++k) {
      jobs.src[j][k] = srcs[j * 6 + k];
      if (k < need && !jobs.src[j][k]) return -4;
      if (k == 0 && kinds[j] == WIMG_MLP_BWD16 && (reinterpret_cast<size_t>(jobs.src[j][k]) & 15)) return -4;   // W3 rows: 16-byte loads
    }
  }
  return 0;
}

}  // namespace
