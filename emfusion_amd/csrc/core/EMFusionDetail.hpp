// EMFusionDetail.hpp -- helpers shared by the translation units of emf::EMFusion (not installed).
#pragma once

#include "EMFusion.hpp"

namespace emf {
namespace detail {

enum Stamp { kStart = 0, kPoints, kEstep, kRaycast, kComposite, kIntegrate, kMasks, kNumStamps };

inline emf_pose_t toPose(const Affine3f& a) {
    emf_pose_t p;
    for (int i = 0; i < 9; ++i) p.R[i] = a.rotation().val[i];
    for (int i = 0; i < 3; ++i) p.t[i] = a.translation().val[i];
    return p;
}

}  // namespace detail
}  // namespace emf
