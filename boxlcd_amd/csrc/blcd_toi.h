// blcd_toi.h — result types of the time-of-impact query (b2TOIOutput, Box2D b2TimeOfImpact.h).
// The query itself lives in blcd_toi_wall.h: in boxLCD every continuous-collision pair is (wall edge, moving shape) - dynamic
// bodies are never bullets (b2World::SolveTOI, reference call site boxLCD/world_env.py:448-450) - so the product carries only
// the wall-specialised, register-resident formulation; the generic b2Distance / b2TimeOfImpact pair exists in the CPU oracle
// alone and is what the parity tests compare this against.
#pragma once
#include "blcd_collide.h"

namespace blcd {

enum TOIState { kTOIUnknown = 0, kTOIFailed, kTOIOverlapped, kTOITouching, kTOISeparated };
struct TOIOutput {
  int state;
  float t;
};

}  // namespace blcd
