// Compile-only check (g++ -fsyntax-only -DMOCK_STRICT_ACCESS): every template of include/orbgpu_dropin.hpp is instantiated against the
// mocks with the reference's access rules switched on -- what is `protected:` in I/MapPoint.h / I/KeyFrame.h / I/Map.h is protected in
// the mocks too.  The glue compiles only if it touches nothing but the reference's public members and the edits INTEGRATION.md lists
// (E1: MapPoint::GetMinDistance / GetMaxDistance, E2: MapPoint::mnChangeStamp, E3: Frame::mpGpuFrame).
// tests/test_reference_access.py runs this and holds the mocks' partition against the reference's headers.
#ifndef MOCK_STRICT_ACCESS
#error "compile with -DMOCK_STRICT_ACCESS"
#endif
#include <set>
#include <vector>

#include "mock_orbslam3.hpp"
#include "orbgpu_dropin.hpp"

namespace od = orbgpu::dropin;
using namespace mock;

struct NoStampOps : od::GpuOps { static constexpr bool kNoChangeStamp = true; static constexpr bool kNoLbaCache = true; };      // an unmodified MapPoint (E1 only)
struct HeuristicOps : od::GpuOps { static constexpr bool kNoChangeStamp = true; static constexpr bool kHeuristicLocalMap = true; };

template int od::isInFrustumAll<od::GpuOps, Frame, MapPoint>(Frame&, const std::vector<MapPoint*>&, float);
template int od::SearchLocalPoints<od::GpuOps, Frame, MapPoint>(Frame&, const std::vector<MapPoint*>&, float, bool, float, float);
template int od::SearchLocalPoints<NoStampOps, Frame, MapPoint>(Frame&, const std::vector<MapPoint*>&, float, bool, float, float);
template int od::SearchLocalPoints<HeuristicOps, Frame, MapPoint>(Frame&, const std::vector<MapPoint*>&, float, bool, float, float);
template int od::SearchByProjection<od::GpuOps, Frame, MapPoint>(Frame&, const std::vector<MapPoint*>&, const float, const bool, const float, float);
template int od::SearchByProjection<od::GpuOps, Frame>(Frame&, const Frame&, const float, const bool, bool);
template int od::SearchByProjection<od::GpuOps, Frame, KeyFrame, MapPoint>(Frame&, KeyFrame*, const std::set<MapPoint*>&, const float, const int, bool);
template int od::SearchByBoW<od::GpuOps, KeyFrame, Frame, MapPoint>(KeyFrame*, Frame&, std::vector<MapPoint*>&, float, bool);
template int od::LocalBundleAdjustment<od::GpuOps, KeyFrame, Map>(KeyFrame*, bool*, Map*, int&, int);
template int od::LocalBundleAdjustment<NoStampOps, KeyFrame, Map>(KeyFrame*, bool*, Map*, int&, int);
template int od::PoseOptimization<od::GpuOps, Frame>(Frame*);
template int od::ComputeStereoFishEyeMatches<od::GpuOps, Frame>(Frame&);
