// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under boxlcd_amd/ may include, link or call this.
//
// CPU restatement of Box2D 2.3.x dynamics for the subset boxLCD drives (reference call sites:
// boxLCD/world_env.py:67,195 b2World; :217-223,249-252,296-302 CreateDynamicBody; :255-267 revoluteJointDef +
// CreateJoint; :311-314 CreateStaticBody(edgeShape); :441 joint.motorSpeed; :448-450 b2World.Step(dt,180,60)).
// Upstream files followed: b2World.cpp, b2Island.cpp, b2ContactManager.cpp, b2Contact.cpp, b2ContactSolver.cpp,
// b2RevoluteJoint.cpp, b2Body.cpp, b2Fixture.cpp, b2BroadPhase.cpp/b2DynamicTree.cpp (pair semantics only: with
// <= 24 proxies the tree is replaced by brute force, which yields the same sorted pair set — SURVEY App. B.3).
//
// This is a general, list-based, one-environment-at-a-time implementation, deliberately organised like Box2D
// (dynamic lists, islands by DFS) and NOT like the HIP product (fixed pair slots, SoA).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>
#include "b2o_toi.h"

namespace b2o {

enum { kStaticBody = 0, kDynamicBody = 2 };
enum { kInactiveLimit = 0, kAtLowerLimit = 1, kAtUpperLimit = 2, kEqualLimits = 3 };

struct Body {
  int type;
  Transform xf;
  Sweep sweep;
  Vec2 v;
  float w;
  float mass, invMass, I, invI;
  float linearDamping, angularDamping;
  float sleepTime;
  bool awake, islandFlag;
  int islandIndex;
  Shape shape;
  float density, friction, restitution;
  uint32_t cat, mask;
  AABB fat;
  std::vector<int> contacts;  // contact-edge list, newest first
  std::vector<int> joints;    // joint-edge list, newest first

  void SetAwake(bool flag) {
    if (flag) {
      if (!awake) {
        awake = true;
        sleepTime = 0.0f;
      }
    } else {
      awake = false;
      sleepTime = 0.0f;
      v = V2(0.0f, 0.0f);
      w = 0.0f;
    }
  }
  void SynchronizeTransform() {
    xf.q.Set(sweep.a);
    xf.p = sweep.c - Mul(xf.q, sweep.localCenter);
  }
  void Advance(float alpha) {
    sweep.Advance(alpha);
    sweep.c = sweep.c0;
    sweep.a = sweep.a0;
    xf.q.Set(sweep.a);
    xf.p = sweep.c - Mul(xf.q, sweep.localCenter);
  }
};

struct Contact {
  bool alive;
  int bodyA, bodyB;
  Manifold m;
  bool enabled, touching, islandFlag, toiFlag;
  int toiCount;
  float toi;
  float friction, restitution;
};

struct Joint {
  int bodyA, bodyB;
  Vec2 localAnchorA, localAnchorB;
  float referenceAngle;
  Vec3 impulse;
  float motorImpulse;
  bool enableLimit, enableMotor;
  float lower, upper, maxMotorTorque, motorSpeed;
  int limitState;
  bool islandFlag;
  // solver temp
  int indexA, indexB;
  Vec2 rA, rB, localCenterA, localCenterB;
  float invMassA, invMassB, invIA, invIB;
  Mat33 mass;
  float motorMass;
};

struct Position {
  Vec2 c;
  float a;
};
struct Velocity {
  Vec2 v;
  float w;
};
struct TimeStep {
  float dt, inv_dt, dtRatio;
  int velocityIterations, positionIterations;
  bool warmStarting;
};

struct VelocityConstraintPoint {
  Vec2 rA, rB;
  float normalImpulse, tangentImpulse, normalMass, tangentMass, velocityBias;
};
struct ContactVelocityConstraint {
  VelocityConstraintPoint points[kMaxManifoldPoints];
  Vec2 normal;
  Mat22 normalMass, K;
  int indexA, indexB;
  float invMassA, invMassB, invIA, invIB, friction, restitution, tangentSpeed;
  int pointCount, contactIndex;
};
struct ContactPositionConstraint {
  Vec2 localPoints[kMaxManifoldPoints];
  Vec2 localNormal, localPoint;
  int indexA, indexB;
  float invMassA, invMassB;
  Vec2 localCenterA, localCenterB;
  float invIA, invIB;
  int type;
  float radiusA, radiusB;
  int pointCount;
};

// diagnostic switches of the sweep trace (B2O_TRACE_NONE=<solves to print>): see World::IslandSolve
inline int& traceLeft() { static int n = std::getenv("B2O_TRACE_NONE") ? std::atoi(std::getenv("B2O_TRACE_NONE")) : 0; return n; }
inline bool traceOn() { static bool on = std::getenv("B2O_TRACE_NONE") != nullptr; return on; }

struct Stats {
  long toiIters = 0;   // diagnostic: minimum-TOI contacts advanced to (true events + those that turn out not to touch)
  long steps = 0, toiEvents = 0, toiCalls = 0, islands = 0, contactsCreated = 0, contactsDestroyed = 0;
  long sweepHist[182] = {0};
  long posFixHist[64] = {0};   // diagnostic: unsolved islands only - iteration at which the positions first repeat (period <= 4), 63 = never
  long posIterHist[62] = {0};  // diagnostic: position iterations used per island solve (61 = not solved within the limit)
  long periodHist[34] = {0};   // diagnostic: period (1..32) of the sweep-state cycle when one is detected, [33] = none, [0] unused
  long cycleAtSum = 0, cycleCount = 0;  // diagnostic: first velocity sweep after which the state is a fixed point (181 = never)
  bool trackSweeps = false;
  long nicHistFree[16] = {0};  // the same for joint-free islands
  long nicHist[16] = {0};      // diagnostic (trackSweeps): contacts per solved island (15 = 15 or more), jointed islands only
  std::vector<int> solveLog;   // diagnostic (trackSweeps): per island solve {fixedAt, cyclePeriod, cycleAt, nJoints, nContacts, nBodies}
  int lastSolveSweeps = 0;     // diagnostic: max over this world step's island solves of the sweeps an exact early exit needs (fixed point, or a cycle of period <= 4 seen within 24 sweeps; else all)
};

struct World;

// b2ContactSolver (b2ContactSolver.cpp)
struct ContactSolver {
  TimeStep step;
  Position* positions;
  Velocity* velocities;
  std::vector<Contact*> contacts;
  std::vector<ContactVelocityConstraint> vcs;
  std::vector<ContactPositionConstraint> pcs;

  void Init(const TimeStep& st, const std::vector<Contact*>& cs, World* w, Position* pos, Velocity* vel);
  void InitializeVelocityConstraints();
  void WarmStart();
  void SolveVelocityConstraints();
  void StoreImpulses();
  bool SolvePositionConstraints();
  bool SolveTOIPositionConstraints(int toiIndexA, int toiIndexB);
};

struct World {
  std::vector<Body> bodies;      // creation order == proxy-id order; Box2D's m_bodyList is this reversed
  std::vector<Contact> contacts; // pool
  std::vector<int> contactList;  // world contact list, newest first
  std::vector<Joint> joints;     // creation order; Box2D's m_jointList is this reversed
  std::vector<int> moveBuffer;
  bool newFixture = false;
  float inv_dt0 = 0.0f;
  Vec2 gravity = V2(0.0f, -9.81f);
  Stats stats;

  // ---- construction (b2World::CreateBody + b2Body::CreateFixture + b2Body::ResetMassData) -----------------
  int CreateBody(int type, Vec2 position, float angle, const Shape& shape, float density, float friction, float restitution,
                 uint32_t cat, uint32_t mask, float linearDamping, float angularDamping) {
    Body b;
    b.type = type;
    b.xf.p = position;
    b.xf.q.Set(angle);
    b.sweep.localCenter = V2(0.0f, 0.0f);
    b.sweep.c0 = b.xf.p;
    b.sweep.c = b.xf.p;
    b.sweep.a0 = angle;
    b.sweep.a = angle;
    b.sweep.alpha0 = 0.0f;
    b.v = V2(0.0f, 0.0f);
    b.w = 0.0f;
    b.linearDamping = linearDamping;
    b.angularDamping = angularDamping;
    b.sleepTime = 0.0f;
    b.awake = true;
    b.islandFlag = false;
    b.islandIndex = 0;
    if (type == kDynamicBody) {
      b.mass = 1.0f;
      b.invMass = 1.0f;
    } else {
      b.mass = 0.0f;
      b.invMass = 0.0f;
    }
    b.I = 0.0f;
    b.invI = 0.0f;
    b.shape = shape;
    b.density = density;
    b.friction = friction;
    b.restitution = restitution;
    b.cat = cat;
    b.mask = mask;
    // CreateFixture -> CreateProxies: fat AABB = tight +- aabbExtension, proxy buffered as moved
    AABB aabb;
    ShapeComputeAABB(&b.shape, &aabb, b.xf);
    Vec2 r = V2(kAabbExtension, kAabbExtension);
    b.fat.lo = aabb.lo - r;
    b.fat.hi = aabb.hi + r;
    bodies.push_back(b);
    int id = (int)bodies.size() - 1;
    moveBuffer.push_back(id);
    newFixture = true;
    if (density > 0.0f) ResetMassData(id);
    return id;
  }

  void ResetMassData(int id) {
    Body& b = bodies[id];
    b.mass = 0.0f;
    b.invMass = 0.0f;
    b.I = 0.0f;
    b.invI = 0.0f;
    b.sweep.localCenter = V2(0.0f, 0.0f);
    if (b.type != kDynamicBody) {
      b.sweep.c0 = b.xf.p;
      b.sweep.c = b.xf.p;
      b.sweep.a0 = b.sweep.a;
      return;
    }
    Vec2 localCenter = V2(0.0f, 0.0f);
    if (b.density != 0.0f) {
      MassData md;
      ShapeComputeMass(&b.shape, &md, b.density);
      b.mass += md.mass;
      localCenter += md.mass * md.center;
      b.I += md.I;
    }
    if (b.mass > 0.0f) {
      b.invMass = 1.0f / b.mass;
      localCenter *= b.invMass;
    } else {
      b.mass = 1.0f;
      b.invMass = 1.0f;
    }
    if (b.I > 0.0f) {
      b.I -= b.mass * Dot(localCenter, localCenter);
      b.invI = 1.0f / b.I;
    } else {
      b.I = 0.0f;
      b.invI = 0.0f;
    }
    Vec2 oldCenter = b.sweep.c;
    b.sweep.localCenter = localCenter;
    b.sweep.c0 = b.sweep.c = Mul(b.xf, b.sweep.localCenter);
    b.v += Cross(b.w, b.sweep.c - oldCenter);
  }

  // b2World::CreateJoint + b2RevoluteJoint ctor (pybox2d's revoluteJointDef(bodyA=,bodyB=,localAnchorA=,...) kwargs path:
  // referenceAngle is taken as bodyB.angle - bodyA.angle at creation — SURVEY App. A / B.9 item 1, [upstream, unpinned])
  int CreateRevoluteJoint(int bodyA, int bodyB, Vec2 anchorA, Vec2 anchorB, bool enableLimit, float lower, float upper,
                          float maxMotorTorque) {
    Joint j;
    j.bodyA = bodyA;
    j.bodyB = bodyB;
    j.localAnchorA = anchorA;
    j.localAnchorB = anchorB;
    j.referenceAngle = bodies[bodyB].sweep.a - bodies[bodyA].sweep.a;
    j.impulse = Vec3{0.0f, 0.0f, 0.0f};
    j.motorImpulse = 0.0f;
    j.enableLimit = enableLimit;
    j.enableMotor = true;
    j.lower = lower;
    j.upper = upper;
    j.maxMotorTorque = maxMotorTorque;
    j.motorSpeed = 0.0f;
    j.limitState = kInactiveLimit;
    j.islandFlag = false;
    joints.push_back(j);
    int id = (int)joints.size() - 1;
    bodies[bodyA].joints.insert(bodies[bodyA].joints.begin(), id);
    bodies[bodyB].joints.insert(bodies[bodyB].joints.begin(), id);
    return id;
  }

  // b2RevoluteJoint::SetMotorSpeed wakes both bodies
  void SetMotorSpeed(int jid, float speed) {
    Joint& j = joints[jid];
    bodies[j.bodyA].SetAwake(true);
    bodies[j.bodyB].SetAwake(true);
    j.motorSpeed = speed;
  }

  // b2Body::SetTransform (2.3.x: re-synchronises proxies with zero displacement, then FindNewContacts)
  void SetTransform(int id, Vec2 position, float angle) {
    Body& b = bodies[id];
    b.xf.q.Set(angle);
    b.xf.p = position;
    b.sweep.c = Mul(b.xf, b.sweep.localCenter);
    b.sweep.a = angle;
    b.sweep.c0 = b.sweep.c;
    b.sweep.a0 = angle;
    SynchronizeProxy(id, b.xf, b.xf);
    FindNewContacts();
  }

  // b2Fixture::Synchronize + b2DynamicTree::MoveProxy
  void SynchronizeProxy(int id, const Transform& xf1, const Transform& xf2) {
    Body& b = bodies[id];
    AABB aabb1, aabb2, aabb;
    ShapeComputeAABB(&b.shape, &aabb1, xf1);
    ShapeComputeAABB(&b.shape, &aabb2, xf2);
    aabb.lo = Min(aabb1.lo, aabb2.lo);
    aabb.hi = Max(aabb1.hi, aabb2.hi);
    Vec2 displacement = xf2.p - xf1.p;
    if (b.fat.Contains(aabb)) return;
    AABB fb = aabb;
    Vec2 r = V2(kAabbExtension, kAabbExtension);
    fb.lo = fb.lo - r;
    fb.hi = fb.hi + r;
    Vec2 d = kAabbMultiplier * displacement;
    if (d.x < 0.0f) fb.lo.x += d.x; else fb.hi.x += d.x;
    if (d.y < 0.0f) fb.lo.y += d.y; else fb.hi.y += d.y;
    b.fat = fb;
    moveBuffer.push_back(id);
  }
  void SynchronizeFixtures(int id) {
    Body& b = bodies[id];
    Transform xf1;
    xf1.q.Set(b.sweep.a0);
    xf1.p = b.sweep.c0 - Mul(xf1.q, b.sweep.localCenter);
    SynchronizeProxy(id, xf1, b.xf);
  }

  // b2Body::ShouldCollide + b2ContactFilter::ShouldCollide
  bool ShouldCollide(int a, int b) const {
    const Body& A = bodies[a];
    const Body& B = bodies[b];
    if (A.type != kDynamicBody && B.type != kDynamicBody) return false;
    for (int jid : B.joints) {
      const Joint& j = joints[jid];
      int other = j.bodyA == b ? j.bodyB : j.bodyA;
      if (other == a) return false;  // collideConnected == false
    }
    return (A.mask & B.cat) != 0 && (A.cat & B.mask) != 0;
  }

  int FindContact(int a, int b) const {
    for (int cid : bodies[b].contacts) {
      const Contact& c = contacts[cid];
      if ((c.bodyA == a && c.bodyB == b) || (c.bodyA == b && c.bodyB == a)) return cid;
    }
    return -1;
  }

  // b2ContactManager::AddPair (proxy a < proxy b)
  void AddPair(int a, int b) {
    if (a == b) return;
    if (FindContact(a, b) >= 0) return;
    if (!ShouldCollide(b, a)) return;
    // b2Contact::Create: s_registers orders (edge,circle) (edge,polygon) (polygon,circle); same-type keeps (a,b)
    int ta = bodies[a].shape.type, tb = bodies[b].shape.type;
    int fa = a, fb = b;
    auto rank = [](int t) { return t == kEdge ? 0 : (t == kPolygon ? 1 : 2); };
    if (rank(ta) > rank(tb)) std::swap(fa, fb);
    Contact c;
    c.alive = true;
    c.bodyA = fa;
    c.bodyB = fb;
    c.m.pointCount = 0;
    c.m.type = 0;
    c.m.localNormal = V2(0.0f, 0.0f);
    c.m.localPoint = V2(0.0f, 0.0f);
    for (int i = 0; i < 2; ++i) {
      c.m.points[i].localPoint = V2(0.0f, 0.0f);
      c.m.points[i].normalImpulse = 0.0f;
      c.m.points[i].tangentImpulse = 0.0f;
      c.m.points[i].id.key = 0;
    }
    c.enabled = true;
    c.touching = false;
    c.islandFlag = false;
    c.toiFlag = false;
    c.toiCount = 0;
    c.toi = 1.0f;
    c.friction = sqrtf(bodies[fa].friction * bodies[fb].friction);
    c.restitution = bodies[fa].restitution > bodies[fb].restitution ? bodies[fa].restitution : bodies[fb].restitution;
    int cid = -1;
    for (size_t i = 0; i < contacts.size(); ++i)
      if (!contacts[i].alive) {
        cid = (int)i;
        break;
      }
    if (cid < 0) {
      contacts.push_back(c);
      cid = (int)contacts.size() - 1;
    } else {
      contacts[cid] = c;
    }
    contactList.insert(contactList.begin(), cid);
    bodies[fa].contacts.insert(bodies[fa].contacts.begin(), cid);
    bodies[fb].contacts.insert(bodies[fb].contacts.begin(), cid);
    bodies[fa].SetAwake(true);
    bodies[fb].SetAwake(true);
    stats.contactsCreated++;
  }

  // b2ContactManager::Destroy + b2Contact::Destroy
  void DestroyContact(int cid) {
    Contact& c = contacts[cid];
    auto rm = [&](std::vector<int>& v) { v.erase(std::remove(v.begin(), v.end(), cid), v.end()); };
    rm(contactList);
    rm(bodies[c.bodyA].contacts);
    rm(bodies[c.bodyB].contacts);
    if (c.m.pointCount > 0) {
      bodies[c.bodyA].SetAwake(true);
      bodies[c.bodyB].SetAwake(true);
    }
    c.alive = false;
    stats.contactsDestroyed++;
  }

  // b2BroadPhase::UpdatePairs + b2ContactManager::FindNewContacts
  void FindNewContacts() {
    std::vector<std::pair<int, int>> pairs;
    for (int p : moveBuffer) {
      for (int q = 0; q < (int)bodies.size(); ++q) {
        if (q == p) continue;
        if (!TestOverlap(bodies[p].fat, bodies[q].fat)) continue;
        pairs.emplace_back(std::min(p, q), std::max(p, q));
      }
    }
    moveBuffer.clear();
    std::sort(pairs.begin(), pairs.end());
    pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
    for (auto& pr : pairs) AddPair(pr.first, pr.second);
  }

  // b2Contact::Evaluate dispatch
  void Evaluate(Contact& c, Manifold* m) {
    const Body& A = bodies[c.bodyA];
    const Body& B = bodies[c.bodyB];
    int ta = A.shape.type, tb = B.shape.type;
    if (ta == kEdge && tb == kCircle) CollideEdgeAndCircle(m, &A.shape, A.xf, &B.shape, B.xf);
    else if (ta == kEdge && tb == kPolygon) CollideEdgeAndPolygon(m, &A.shape, A.xf, &B.shape, B.xf);
    else if (ta == kPolygon && tb == kCircle) CollidePolygonAndCircle(m, &A.shape, A.xf, &B.shape, B.xf);
    else if (ta == kPolygon && tb == kPolygon) CollidePolygons(m, &A.shape, A.xf, &B.shape, B.xf);
    else CollideCircles(m, &A.shape, A.xf, &B.shape, B.xf);
  }

  // b2Contact::Update
  void UpdateContact(int cid) {
    Contact& c = contacts[cid];
    Manifold oldManifold = c.m;
    c.enabled = true;
    bool wasTouching = c.touching;
    Evaluate(c, &c.m);
    bool touching = c.m.pointCount > 0;
    for (int i = 0; i < c.m.pointCount; ++i) {
      ManifoldPoint* mp2 = c.m.points + i;
      mp2->normalImpulse = 0.0f;
      mp2->tangentImpulse = 0.0f;
      ContactID id2 = mp2->id;
      for (int j = 0; j < oldManifold.pointCount; ++j) {
        ManifoldPoint* mp1 = oldManifold.points + j;
        if (mp1->id.key == id2.key) {
          mp2->normalImpulse = mp1->normalImpulse;
          mp2->tangentImpulse = mp1->tangentImpulse;
          break;
        }
      }
    }
    if (touching != wasTouching) {
      bodies[c.bodyA].SetAwake(true);
      bodies[c.bodyB].SetAwake(true);
    }
    c.touching = touching;
  }

  // b2ContactManager::Collide
  void Collide() {
    std::vector<int> list = contactList;  // iteration order fixed at entry; destruction only removes the current one
    for (int cid : list) {
      Contact& c = contacts[cid];
      Body& A = bodies[c.bodyA];
      Body& B = bodies[c.bodyB];
      bool activeA = A.awake && A.type != kStaticBody;
      bool activeB = B.awake && B.type != kStaticBody;
      if (!activeA && !activeB) continue;
      if (!TestOverlap(A.fat, B.fat)) {
        DestroyContact(cid);
        continue;
      }
      UpdateContact(cid);
    }
  }

  void Solve(const TimeStep& step);
  void SolveTOI(const TimeStep& step);
  void IslandSolve(std::vector<int>& ibodies, std::vector<int>& icontacts, std::vector<int>& ijoints, const TimeStep& step);
  void IslandSolveTOI(std::vector<int>& ibodies, std::vector<int>& icontacts, const TimeStep& subStep, int toiIndexA, int toiIndexB);

  // b2World::Step
  void Step(float dt, int velocityIterations, int positionIterations) {
    if (newFixture) {
      FindNewContacts();
      newFixture = false;
    }
    TimeStep step;
    step.dt = dt;
    step.velocityIterations = velocityIterations;
    step.positionIterations = positionIterations;
    if (dt > 0.0f) step.inv_dt = 1.0f / dt; else step.inv_dt = 0.0f;
    step.dtRatio = inv_dt0 * dt;
    step.warmStarting = true;
    Collide();
    if (step.dt > 0.0f) Solve(step);
    if (step.dt > 0.0f) SolveTOI(step);
    if (step.dt > 0.0f) inv_dt0 = step.inv_dt;
    stats.steps++;
  }
};

// ------------------------------------------------------------------------------------------------------
// b2RevoluteJoint (b2RevoluteJoint.cpp)
// ------------------------------------------------------------------------------------------------------
static inline void JointInitVelocityConstraints(Joint& j, World& w, const TimeStep& step, Position* positions, Velocity* velocities) {
  Body& bA = w.bodies[j.bodyA];
  Body& bB = w.bodies[j.bodyB];
  j.indexA = bA.islandIndex;
  j.indexB = bB.islandIndex;
  j.localCenterA = bA.sweep.localCenter;
  j.localCenterB = bB.sweep.localCenter;
  j.invMassA = bA.invMass;
  j.invMassB = bB.invMass;
  j.invIA = bA.invI;
  j.invIB = bB.invI;
  float aA = positions[j.indexA].a;
  Vec2 vA = velocities[j.indexA].v;
  float wA = velocities[j.indexA].w;
  float aB = positions[j.indexB].a;
  Vec2 vB = velocities[j.indexB].v;
  float wB = velocities[j.indexB].w;
  Rot qA = MakeRot(aA), qB = MakeRot(aB);
  j.rA = Mul(qA, j.localAnchorA - j.localCenterA);
  j.rB = Mul(qB, j.localAnchorB - j.localCenterB);
  float mA = j.invMassA, mB = j.invMassB;
  float iA = j.invIA, iB = j.invIB;
  bool fixedRotation = (iA + iB == 0.0f);
  j.mass.ex.x = mA + mB + j.rA.y * j.rA.y * iA + j.rB.y * j.rB.y * iB;
  j.mass.ey.x = -j.rA.y * j.rA.x * iA - j.rB.y * j.rB.x * iB;
  j.mass.ez.x = -j.rA.y * iA - j.rB.y * iB;
  j.mass.ex.y = j.mass.ey.x;
  j.mass.ey.y = mA + mB + j.rA.x * j.rA.x * iA + j.rB.x * j.rB.x * iB;
  j.mass.ez.y = j.rA.x * iA + j.rB.x * iB;
  j.mass.ex.z = j.mass.ez.x;
  j.mass.ey.z = j.mass.ez.y;
  j.mass.ez.z = iA + iB;
  j.motorMass = iA + iB;
  if (j.motorMass > 0.0f) j.motorMass = 1.0f / j.motorMass;
  if (j.enableMotor == false || fixedRotation) j.motorImpulse = 0.0f;
  if (j.enableLimit && fixedRotation == false) {
    float jointAngle = aB - aA - j.referenceAngle;
    if (Abs(j.upper - j.lower) < 2.0f * kAngularSlop) {
      j.limitState = kEqualLimits;
    } else if (jointAngle <= j.lower) {
      if (j.limitState != kAtLowerLimit) j.impulse.z = 0.0f;
      j.limitState = kAtLowerLimit;
    } else if (jointAngle >= j.upper) {
      if (j.limitState != kAtUpperLimit) j.impulse.z = 0.0f;
      j.limitState = kAtUpperLimit;
    } else {
      j.limitState = kInactiveLimit;
      j.impulse.z = 0.0f;
    }
  } else {
    j.limitState = kInactiveLimit;
  }
  if (step.warmStarting) {
    j.impulse *= step.dtRatio;
    j.motorImpulse *= step.dtRatio;
    Vec2 P = V2(j.impulse.x, j.impulse.y);
    vA -= mA * P;
    wA -= iA * (Cross(j.rA, P) + j.motorImpulse + j.impulse.z);
    vB += mB * P;
    wB += iB * (Cross(j.rB, P) + j.motorImpulse + j.impulse.z);
  } else {
    j.impulse = Vec3{0.0f, 0.0f, 0.0f};
    j.motorImpulse = 0.0f;
  }
  velocities[j.indexA].v = vA;
  velocities[j.indexA].w = wA;
  velocities[j.indexB].v = vB;
  velocities[j.indexB].w = wB;
}

static inline void JointSolveVelocityConstraints(Joint& j, const TimeStep& step, Velocity* velocities) {
  Vec2 vA = velocities[j.indexA].v;
  float wA = velocities[j.indexA].w;
  Vec2 vB = velocities[j.indexB].v;
  float wB = velocities[j.indexB].w;
  float mA = j.invMassA, mB = j.invMassB;
  float iA = j.invIA, iB = j.invIB;
  bool fixedRotation = (iA + iB == 0.0f);
  if (j.enableMotor && j.limitState != kEqualLimits && fixedRotation == false) {
    float Cdot = wB - wA - j.motorSpeed;
    float impulse = -j.motorMass * Cdot;
    float oldImpulse = j.motorImpulse;
    float maxImpulse = step.dt * j.maxMotorTorque;
    j.motorImpulse = Clamp(j.motorImpulse + impulse, -maxImpulse, maxImpulse);
    impulse = j.motorImpulse - oldImpulse;
    wA -= iA * impulse;
    wB += iB * impulse;
  }
  if (j.enableLimit && j.limitState != kInactiveLimit && fixedRotation == false) {
    Vec2 Cdot1 = vB + Cross(wB, j.rB) - vA - Cross(wA, j.rA);
    float Cdot2 = wB - wA;
    Vec3 Cdot = Vec3{Cdot1.x, Cdot1.y, Cdot2};
    Vec3 impulse = -j.mass.Solve33(Cdot);
    if (j.limitState == kEqualLimits) {
      j.impulse += impulse;
    } else if (j.limitState == kAtLowerLimit) {
      float newImpulse = j.impulse.z + impulse.z;
      if (newImpulse < 0.0f) {
        Vec2 rhs = -Cdot1 + j.impulse.z * V2(j.mass.ez.x, j.mass.ez.y);
        Vec2 reduced = j.mass.Solve22(rhs);
        impulse.x = reduced.x;
        impulse.y = reduced.y;
        impulse.z = -j.impulse.z;
        j.impulse.x += reduced.x;
        j.impulse.y += reduced.y;
        j.impulse.z = 0.0f;
      } else {
        j.impulse += impulse;
      }
    } else if (j.limitState == kAtUpperLimit) {
      float newImpulse = j.impulse.z + impulse.z;
      if (newImpulse > 0.0f) {
        Vec2 rhs = -Cdot1 + j.impulse.z * V2(j.mass.ez.x, j.mass.ez.y);
        Vec2 reduced = j.mass.Solve22(rhs);
        impulse.x = reduced.x;
        impulse.y = reduced.y;
        impulse.z = -j.impulse.z;
        j.impulse.x += reduced.x;
        j.impulse.y += reduced.y;
        j.impulse.z = 0.0f;
      } else {
        j.impulse += impulse;
      }
    }
    Vec2 P = V2(impulse.x, impulse.y);
    vA -= mA * P;
    wA -= iA * (Cross(j.rA, P) + impulse.z);
    vB += mB * P;
    wB += iB * (Cross(j.rB, P) + impulse.z);
  } else {
    Vec2 Cdot = vB + Cross(wB, j.rB) - vA - Cross(wA, j.rA);
    Vec2 impulse = j.mass.Solve22(-Cdot);
    j.impulse.x += impulse.x;
    j.impulse.y += impulse.y;
    vA -= mA * impulse;
    wA -= iA * Cross(j.rA, impulse);
    vB += mB * impulse;
    wB += iB * Cross(j.rB, impulse);
  }
  velocities[j.indexA].v = vA;
  velocities[j.indexA].w = wA;
  velocities[j.indexB].v = vB;
  velocities[j.indexB].w = wB;
}

static inline bool JointSolvePositionConstraints(Joint& j, Position* positions) {
  Vec2 cA = positions[j.indexA].c;
  float aA = positions[j.indexA].a;
  Vec2 cB = positions[j.indexB].c;
  float aB = positions[j.indexB].a;
  Rot qA, qB;
  float angularError = 0.0f;
  float positionError = 0.0f;
  bool fixedRotation = (j.invIA + j.invIB == 0.0f);
  if (j.enableLimit && j.limitState != kInactiveLimit && fixedRotation == false) {
    float angle = aB - aA - j.referenceAngle;
    float limitImpulse = 0.0f;
    if (j.limitState == kEqualLimits) {
      float C = Clamp(angle - j.lower, -kMaxAngularCorrection, kMaxAngularCorrection);
      limitImpulse = -j.motorMass * C;
      angularError = Abs(C);
    } else if (j.limitState == kAtLowerLimit) {
      float C = angle - j.lower;
      angularError = -C;
      C = Clamp(C + kAngularSlop, -kMaxAngularCorrection, 0.0f);
      limitImpulse = -j.motorMass * C;
    } else if (j.limitState == kAtUpperLimit) {
      float C = angle - j.upper;
      angularError = C;
      C = Clamp(C - kAngularSlop, 0.0f, kMaxAngularCorrection);
      limitImpulse = -j.motorMass * C;
    }
    aA -= j.invIA * limitImpulse;
    aB += j.invIB * limitImpulse;
  }
  {
    qA.Set(aA);
    qB.Set(aB);
    Vec2 rA = Mul(qA, j.localAnchorA - j.localCenterA);
    Vec2 rB = Mul(qB, j.localAnchorB - j.localCenterB);
    Vec2 C = cB + rB - cA - rA;
    positionError = Length(C);
    float mA = j.invMassA, mB = j.invMassB;
    float iA = j.invIA, iB = j.invIB;
    Mat22 K;
    K.ex.x = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y;
    K.ex.y = -iA * rA.x * rA.y - iB * rB.x * rB.y;
    K.ey.x = K.ex.y;
    K.ey.y = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
    Vec2 impulse = -K.Solve(C);
    cA -= mA * impulse;
    aA -= iA * Cross(rA, impulse);
    cB += mB * impulse;
    aB += iB * Cross(rB, impulse);
  }
  positions[j.indexA].c = cA;
  positions[j.indexA].a = aA;
  positions[j.indexB].c = cB;
  positions[j.indexB].a = aB;
  return positionError <= kLinearSlop && angularError <= kAngularSlop;
}

// ------------------------------------------------------------------------------------------------------
// b2ContactSolver
// ------------------------------------------------------------------------------------------------------
inline void ContactSolver::Init(const TimeStep& st, const std::vector<Contact*>& cs, World* w, Position* pos, Velocity* vel) {
  step = st;
  positions = pos;
  velocities = vel;
  contacts = cs;
  vcs.assign(cs.size(), ContactVelocityConstraint());
  pcs.assign(cs.size(), ContactPositionConstraint());
  for (size_t i = 0; i < cs.size(); ++i) {
    Contact* contact = cs[i];
    Body& bodyA = w->bodies[contact->bodyA];
    Body& bodyB = w->bodies[contact->bodyB];
    float radiusA = bodyA.shape.radius;
    float radiusB = bodyB.shape.radius;
    Manifold* manifold = &contact->m;
    int pointCount = manifold->pointCount;
    ContactVelocityConstraint* vc = &vcs[i];
    vc->friction = contact->friction;
    vc->restitution = contact->restitution;
    vc->tangentSpeed = 0.0f;
    vc->indexA = bodyA.islandIndex;
    vc->indexB = bodyB.islandIndex;
    vc->invMassA = bodyA.invMass;
    vc->invMassB = bodyB.invMass;
    vc->invIA = bodyA.invI;
    vc->invIB = bodyB.invI;
    vc->contactIndex = (int)i;
    vc->pointCount = pointCount;
    vc->K.ex = vc->K.ey = V2(0.0f, 0.0f);
    vc->normalMass.ex = vc->normalMass.ey = V2(0.0f, 0.0f);
    ContactPositionConstraint* pc = &pcs[i];
    pc->indexA = bodyA.islandIndex;
    pc->indexB = bodyB.islandIndex;
    pc->invMassA = bodyA.invMass;
    pc->invMassB = bodyB.invMass;
    pc->localCenterA = bodyA.sweep.localCenter;
    pc->localCenterB = bodyB.sweep.localCenter;
    pc->invIA = bodyA.invI;
    pc->invIB = bodyB.invI;
    pc->localNormal = manifold->localNormal;
    pc->localPoint = manifold->localPoint;
    pc->pointCount = pointCount;
    pc->radiusA = radiusA;
    pc->radiusB = radiusB;
    pc->type = manifold->type;
    for (int jx = 0; jx < pointCount; ++jx) {
      ManifoldPoint* cp = manifold->points + jx;
      VelocityConstraintPoint* vcp = vc->points + jx;
      if (step.warmStarting) {
        vcp->normalImpulse = step.dtRatio * cp->normalImpulse;
        vcp->tangentImpulse = step.dtRatio * cp->tangentImpulse;
      } else {
        vcp->normalImpulse = 0.0f;
        vcp->tangentImpulse = 0.0f;
      }
      vcp->rA = V2(0.0f, 0.0f);
      vcp->rB = V2(0.0f, 0.0f);
      vcp->normalMass = 0.0f;
      vcp->tangentMass = 0.0f;
      vcp->velocityBias = 0.0f;
      pc->localPoints[jx] = cp->localPoint;
    }
  }
}

inline void ContactSolver::InitializeVelocityConstraints() {
  for (size_t i = 0; i < vcs.size(); ++i) {
    ContactVelocityConstraint* vc = &vcs[i];
    ContactPositionConstraint* pc = &pcs[i];
    float radiusA = pc->radiusA, radiusB = pc->radiusB;
    Manifold* manifold = &contacts[vc->contactIndex]->m;
    int indexA = vc->indexA, indexB = vc->indexB;
    float mA = vc->invMassA, mB = vc->invMassB;
    float iA = vc->invIA, iB = vc->invIB;
    Vec2 localCenterA = pc->localCenterA, localCenterB = pc->localCenterB;
    Vec2 cA = positions[indexA].c;
    float aA = positions[indexA].a;
    Vec2 vA = velocities[indexA].v;
    float wA = velocities[indexA].w;
    Vec2 cB = positions[indexB].c;
    float aB = positions[indexB].a;
    Vec2 vB = velocities[indexB].v;
    float wB = velocities[indexB].w;
    Transform xfA, xfB;
    xfA.q.Set(aA);
    xfB.q.Set(aB);
    xfA.p = cA - Mul(xfA.q, localCenterA);
    xfB.p = cB - Mul(xfB.q, localCenterB);
    WorldManifold worldManifold;
    worldManifold.Initialize(manifold, xfA, radiusA, xfB, radiusB);
    vc->normal = worldManifold.normal;
    int pointCount = vc->pointCount;
    for (int j = 0; j < pointCount; ++j) {
      VelocityConstraintPoint* vcp = vc->points + j;
      vcp->rA = worldManifold.points[j] - cA;
      vcp->rB = worldManifold.points[j] - cB;
      float rnA = Cross(vcp->rA, vc->normal);
      float rnB = Cross(vcp->rB, vc->normal);
      float kNormal = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
      vcp->normalMass = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
      Vec2 tangent = Cross(vc->normal, 1.0f);
      float rtA = Cross(vcp->rA, tangent);
      float rtB = Cross(vcp->rB, tangent);
      float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
      vcp->tangentMass = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
      vcp->velocityBias = 0.0f;
      float vRel = Dot(vc->normal, vB + Cross(wB, vcp->rB) - vA - Cross(wA, vcp->rA));
      if (vRel < -kVelocityThreshold) vcp->velocityBias = -vc->restitution * vRel;
    }
    if (vc->pointCount == 2) {
      VelocityConstraintPoint* vcp1 = vc->points + 0;
      VelocityConstraintPoint* vcp2 = vc->points + 1;
      float rn1A = Cross(vcp1->rA, vc->normal);
      float rn1B = Cross(vcp1->rB, vc->normal);
      float rn2A = Cross(vcp2->rA, vc->normal);
      float rn2B = Cross(vcp2->rB, vc->normal);
      float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
      float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
      float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
      const float k_maxConditionNumber = 1000.0f;
      if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12)) {
        vc->K.ex = V2(k11, k12);
        vc->K.ey = V2(k12, k22);
        vc->normalMass = vc->K.GetInverse();
      } else {
        vc->pointCount = 1;
      }
    }
  }
}

inline void ContactSolver::WarmStart() {
  for (size_t i = 0; i < vcs.size(); ++i) {
    ContactVelocityConstraint* vc = &vcs[i];
    int indexA = vc->indexA, indexB = vc->indexB;
    float mA = vc->invMassA, iA = vc->invIA, mB = vc->invMassB, iB = vc->invIB;
    int pointCount = vc->pointCount;
    Vec2 vA = velocities[indexA].v;
    float wA = velocities[indexA].w;
    Vec2 vB = velocities[indexB].v;
    float wB = velocities[indexB].w;
    Vec2 normal = vc->normal;
    Vec2 tangent = Cross(normal, 1.0f);
    for (int j = 0; j < pointCount; ++j) {
      VelocityConstraintPoint* vcp = vc->points + j;
      Vec2 P = vcp->normalImpulse * normal + vcp->tangentImpulse * tangent;
      wA -= iA * Cross(vcp->rA, P);
      vA -= mA * P;
      wB += iB * Cross(vcp->rB, P);
      vB += mB * P;
    }
    velocities[indexA].v = vA;
    velocities[indexA].w = wA;
    velocities[indexB].v = vB;
    velocities[indexB].w = wB;
  }
}

inline void ContactSolver::SolveVelocityConstraints() {
  for (size_t i = 0; i < vcs.size(); ++i) {
    ContactVelocityConstraint* vc = &vcs[i];
    int indexA = vc->indexA, indexB = vc->indexB;
    float mA = vc->invMassA, iA = vc->invIA, mB = vc->invMassB, iB = vc->invIB;
    int pointCount = vc->pointCount;
    Vec2 vA = velocities[indexA].v;
    float wA = velocities[indexA].w;
    Vec2 vB = velocities[indexB].v;
    float wB = velocities[indexB].w;
    Vec2 normal = vc->normal;
    Vec2 tangent = Cross(normal, 1.0f);
    float friction = vc->friction;
    for (int j = 0; j < pointCount; ++j) {
      VelocityConstraintPoint* vcp = vc->points + j;
      Vec2 dv = vB + Cross(wB, vcp->rB) - vA - Cross(wA, vcp->rA);
      float vt = Dot(dv, tangent) - vc->tangentSpeed;
      float lambda = vcp->tangentMass * (-vt);
      float maxFriction = friction * vcp->normalImpulse;
      float newImpulse = Clamp(vcp->tangentImpulse + lambda, -maxFriction, maxFriction);
      lambda = newImpulse - vcp->tangentImpulse;
      vcp->tangentImpulse = newImpulse;
      Vec2 P = lambda * tangent;
      vA -= mA * P;
      wA -= iA * Cross(vcp->rA, P);
      vB += mB * P;
      wB += iB * Cross(vcp->rB, P);
    }
    if (vc->pointCount == 1) {
      VelocityConstraintPoint* vcp = vc->points + 0;
      Vec2 dv = vB + Cross(wB, vcp->rB) - vA - Cross(wA, vcp->rA);
      float vn = Dot(dv, normal);
      float lambda = -vcp->normalMass * (vn - vcp->velocityBias);
      float newImpulse = Max(vcp->normalImpulse + lambda, 0.0f);
      lambda = newImpulse - vcp->normalImpulse;
      vcp->normalImpulse = newImpulse;
      Vec2 P = lambda * normal;
      vA -= mA * P;
      wA -= iA * Cross(vcp->rA, P);
      vB += mB * P;
      wB += iB * Cross(vcp->rB, P);
    } else {
      VelocityConstraintPoint* cp1 = vc->points + 0;
      VelocityConstraintPoint* cp2 = vc->points + 1;
      Vec2 a = V2(cp1->normalImpulse, cp2->normalImpulse);
      Vec2 dv1 = vB + Cross(wB, cp1->rB) - vA - Cross(wA, cp1->rA);
      Vec2 dv2 = vB + Cross(wB, cp2->rB) - vA - Cross(wA, cp2->rA);
      float vn1 = Dot(dv1, normal);
      float vn2 = Dot(dv2, normal);
      Vec2 b;
      b.x = vn1 - cp1->velocityBias;
      b.y = vn2 - cp2->velocityBias;
      b -= Mul(vc->K, a);
      for (;;) {
        Vec2 x = -Mul(vc->normalMass, b);
        if (x.x >= 0.0f && x.y >= 0.0f) {
          Vec2 d = x - a;
          Vec2 P1 = d.x * normal;
          Vec2 P2 = d.y * normal;
          vA -= mA * (P1 + P2);
          wA -= iA * (Cross(cp1->rA, P1) + Cross(cp2->rA, P2));
          vB += mB * (P1 + P2);
          wB += iB * (Cross(cp1->rB, P1) + Cross(cp2->rB, P2));
          cp1->normalImpulse = x.x;
          cp2->normalImpulse = x.y;
          break;
        }
        x.x = -cp1->normalMass * b.x;
        x.y = 0.0f;
        vn1 = 0.0f;
        vn2 = vc->K.ex.y * x.x + b.y;
        if (x.x >= 0.0f && vn2 >= 0.0f) {
          Vec2 d = x - a;
          Vec2 P1 = d.x * normal;
          Vec2 P2 = d.y * normal;
          vA -= mA * (P1 + P2);
          wA -= iA * (Cross(cp1->rA, P1) + Cross(cp2->rA, P2));
          vB += mB * (P1 + P2);
          wB += iB * (Cross(cp1->rB, P1) + Cross(cp2->rB, P2));
          cp1->normalImpulse = x.x;
          cp2->normalImpulse = x.y;
          break;
        }
        x.x = 0.0f;
        x.y = -cp2->normalMass * b.y;
        vn1 = vc->K.ey.x * x.y + b.x;
        vn2 = 0.0f;
        if (x.y >= 0.0f && vn1 >= 0.0f) {
          Vec2 d = x - a;
          Vec2 P1 = d.x * normal;
          Vec2 P2 = d.y * normal;
          vA -= mA * (P1 + P2);
          wA -= iA * (Cross(cp1->rA, P1) + Cross(cp2->rA, P2));
          vB += mB * (P1 + P2);
          wB += iB * (Cross(cp1->rB, P1) + Cross(cp2->rB, P2));
          cp1->normalImpulse = x.x;
          cp2->normalImpulse = x.y;
          break;
        }
        x.x = 0.0f;
        x.y = 0.0f;
        vn1 = b.x;
        vn2 = b.y;
        if (vn1 >= 0.0f && vn2 >= 0.0f) {
          Vec2 d = x - a;
          Vec2 P1 = d.x * normal;
          Vec2 P2 = d.y * normal;
          vA -= mA * (P1 + P2);
          wA -= iA * (Cross(cp1->rA, P1) + Cross(cp2->rA, P2));
          vB += mB * (P1 + P2);
          wB += iB * (Cross(cp1->rB, P1) + Cross(cp2->rB, P2));
          cp1->normalImpulse = x.x;
          cp2->normalImpulse = x.y;
          break;
        }
        break;
      }
    }
    velocities[indexA].v = vA;
    velocities[indexA].w = wA;
    velocities[indexB].v = vB;
    velocities[indexB].w = wB;
  }
}

inline void ContactSolver::StoreImpulses() {
  for (size_t i = 0; i < vcs.size(); ++i) {
    ContactVelocityConstraint* vc = &vcs[i];
    Manifold* manifold = &contacts[vc->contactIndex]->m;
    for (int j = 0; j < vc->pointCount; ++j) {
      manifold->points[j].normalImpulse = vc->points[j].normalImpulse;
      manifold->points[j].tangentImpulse = vc->points[j].tangentImpulse;
    }
  }
}

struct PositionSolverManifold {
  Vec2 normal, point;
  float separation;
  void Initialize(const ContactPositionConstraint* pc, const Transform& xfA, const Transform& xfB, int index) {
    switch (pc->type) {
      case kManifoldCircles: {
        Vec2 pointA = Mul(xfA, pc->localPoint);
        Vec2 pointB = Mul(xfB, pc->localPoints[0]);
        normal = pointB - pointA;
        Normalize(normal);
        point = 0.5f * (pointA + pointB);
        separation = Dot(pointB - pointA, normal) - pc->radiusA - pc->radiusB;
      } break;
      case kManifoldFaceA: {
        normal = Mul(xfA.q, pc->localNormal);
        Vec2 planePoint = Mul(xfA, pc->localPoint);
        Vec2 clipPoint = Mul(xfB, pc->localPoints[index]);
        separation = Dot(clipPoint - planePoint, normal) - pc->radiusA - pc->radiusB;
        point = clipPoint;
      } break;
      case kManifoldFaceB: {
        normal = Mul(xfB.q, pc->localNormal);
        Vec2 planePoint = Mul(xfB, pc->localPoint);
        Vec2 clipPoint = Mul(xfA, pc->localPoints[index]);
        separation = Dot(clipPoint - planePoint, normal) - pc->radiusA - pc->radiusB;
        point = clipPoint;
        normal = -normal;
      } break;
    }
  }
};

inline bool ContactSolver::SolvePositionConstraints() {
  float minSeparation = 0.0f;
  for (size_t i = 0; i < pcs.size(); ++i) {
    ContactPositionConstraint* pc = &pcs[i];
    int indexA = pc->indexA, indexB = pc->indexB;
    Vec2 localCenterA = pc->localCenterA;
    float mA = pc->invMassA, iA = pc->invIA;
    Vec2 localCenterB = pc->localCenterB;
    float mB = pc->invMassB, iB = pc->invIB;
    int pointCount = pc->pointCount;
    Vec2 cA = positions[indexA].c;
    float aA = positions[indexA].a;
    Vec2 cB = positions[indexB].c;
    float aB = positions[indexB].a;
    for (int j = 0; j < pointCount; ++j) {
      Transform xfA, xfB;
      xfA.q.Set(aA);
      xfB.q.Set(aB);
      xfA.p = cA - Mul(xfA.q, localCenterA);
      xfB.p = cB - Mul(xfB.q, localCenterB);
      PositionSolverManifold psm;
      psm.Initialize(pc, xfA, xfB, j);
      Vec2 normal = psm.normal;
      Vec2 point = psm.point;
      float separation = psm.separation;
      Vec2 rA = point - cA;
      Vec2 rB = point - cB;
      minSeparation = Min(minSeparation, separation);
      float C = Clamp(kBaumgarte * (separation + kLinearSlop), -kMaxLinearCorrection, 0.0f);
      float rnA = Cross(rA, normal);
      float rnB = Cross(rB, normal);
      float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
      float impulse = K > 0.0f ? -C / K : 0.0f;
      Vec2 P = impulse * normal;
      cA -= mA * P;
      aA -= iA * Cross(rA, P);
      cB += mB * P;
      aB += iB * Cross(rB, P);
    }
    positions[indexA].c = cA;
    positions[indexA].a = aA;
    positions[indexB].c = cB;
    positions[indexB].a = aB;
  }
  return minSeparation >= -3.0f * kLinearSlop;
}

inline bool ContactSolver::SolveTOIPositionConstraints(int toiIndexA, int toiIndexB) {
  float minSeparation = 0.0f;
  for (size_t i = 0; i < pcs.size(); ++i) {
    ContactPositionConstraint* pc = &pcs[i];
    int indexA = pc->indexA, indexB = pc->indexB;
    Vec2 localCenterA = pc->localCenterA;
    Vec2 localCenterB = pc->localCenterB;
    int pointCount = pc->pointCount;
    float mA = 0.0f, iA = 0.0f;
    if (indexA == toiIndexA || indexA == toiIndexB) {
      mA = pc->invMassA;
      iA = pc->invIA;
    }
    float mB = 0.0f, iB = 0.0f;
    if (indexB == toiIndexA || indexB == toiIndexB) {
      mB = pc->invMassB;
      iB = pc->invIB;
    }
    Vec2 cA = positions[indexA].c;
    float aA = positions[indexA].a;
    Vec2 cB = positions[indexB].c;
    float aB = positions[indexB].a;
    for (int j = 0; j < pointCount; ++j) {
      Transform xfA, xfB;
      xfA.q.Set(aA);
      xfB.q.Set(aB);
      xfA.p = cA - Mul(xfA.q, localCenterA);
      xfB.p = cB - Mul(xfB.q, localCenterB);
      PositionSolverManifold psm;
      psm.Initialize(pc, xfA, xfB, j);
      Vec2 normal = psm.normal;
      Vec2 point = psm.point;
      float separation = psm.separation;
      Vec2 rA = point - cA;
      Vec2 rB = point - cB;
      minSeparation = Min(minSeparation, separation);
      float C = Clamp(kToiBaumgarte * (separation + kLinearSlop), -kMaxLinearCorrection, 0.0f);
      float rnA = Cross(rA, normal);
      float rnB = Cross(rB, normal);
      float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
      float impulse = K > 0.0f ? -C / K : 0.0f;
      Vec2 P = impulse * normal;
      cA -= mA * P;
      aA -= iA * Cross(rA, P);
      cB += mB * P;
      aB += iB * Cross(rB, P);
    }
    positions[indexA].c = cA;
    positions[indexA].a = aA;
    positions[indexB].c = cB;
    positions[indexB].a = aB;
  }
  return minSeparation >= -1.5f * kLinearSlop;
}

// ------------------------------------------------------------------------------------------------------
// b2Island::Solve (b2Island.cpp)
// ------------------------------------------------------------------------------------------------------
inline void World::IslandSolve(std::vector<int>& ibodies, std::vector<int>& icontacts, std::vector<int>& ijoints,
                               const TimeStep& step) {
  float h = step.dt;
  int bodyCount = (int)ibodies.size();
  std::vector<Position> positions(bodyCount);
  std::vector<Velocity> velocities(bodyCount);
  for (int i = 0; i < bodyCount; ++i) {
    Body& b = bodies[ibodies[i]];
    Vec2 c = b.sweep.c;
    float a = b.sweep.a;
    Vec2 v = b.v;
    float w = b.w;
    b.sweep.c0 = b.sweep.c;
    b.sweep.a0 = b.sweep.a;
    if (b.type == kDynamicBody) {
      v += h * (1.0f * gravity + b.invMass * V2(0.0f, 0.0f));
      w += h * b.invI * 0.0f;
      if (b2o_variant(1) == 1) {  // Box2D 2.3.0 / 2.2.x: first-order Taylor form
        v *= Clamp(1.0f - h * b.linearDamping, 0.0f, 1.0f);
        w *= Clamp(1.0f - h * b.angularDamping, 0.0f, 1.0f);
      } else {                    // Box2D >= 2.3.1: Pade form
        v *= 1.0f / (1.0f + h * b.linearDamping);
        w *= 1.0f / (1.0f + h * b.angularDamping);
      }
    }
    positions[i].c = c;
    positions[i].a = a;
    velocities[i].v = v;
    velocities[i].w = w;
  }
  std::vector<Contact*> cs;
  for (int cid : icontacts) cs.push_back(&contacts[cid]);
  ContactSolver contactSolver;
  contactSolver.Init(step, cs, this, positions.data(), velocities.data());
  contactSolver.InitializeVelocityConstraints();
  if (step.warmStarting) contactSolver.WarmStart();
  for (int jid : ijoints) JointInitVelocityConstraints(joints[jid], *this, step, positions.data(), velocities.data());
  int fixedAt = -1;
  std::vector<unsigned long long> hashes;
  int cyclePeriod = 0, cycleAt = -1;
  std::vector<std::vector<float>> trace;
  for (int i = 0; i < step.velocityIterations; ++i) {
    std::vector<Velocity> v0;
    std::vector<ContactVelocityConstraint> c0;
    std::vector<Joint> j0;
    if (stats.trackSweeps && fixedAt < 0) {
      v0 = velocities;
      c0 = contactSolver.vcs;
      for (int jid : ijoints) j0.push_back(joints[jid]);
    }
    for (int jid : ijoints) JointSolveVelocityConstraints(joints[jid], step, velocities.data());
    contactSolver.SolveVelocityConstraints();
    if (stats.trackSweeps && fixedAt < 0) {
      bool same = true;
      for (size_t k = 0; k < v0.size(); ++k)
        same = same && v0[k].v.x == velocities[k].v.x && v0[k].v.y == velocities[k].v.y && v0[k].w == velocities[k].w;
      for (size_t k = 0; k < c0.size(); ++k)
        for (int q = 0; q < 2; ++q)
          same = same && c0[k].points[q].normalImpulse == contactSolver.vcs[k].points[q].normalImpulse &&
                 c0[k].points[q].tangentImpulse == contactSolver.vcs[k].points[q].tangentImpulse;
      for (size_t k = 0; k < j0.size(); ++k) {
        const Joint& jj = joints[ijoints[k]];
        same = same && j0[k].impulse.x == jj.impulse.x && j0[k].impulse.y == jj.impulse.y && j0[k].impulse.z == jj.impulse.z &&
               j0[k].motorImpulse == jj.motorImpulse;
      }
      if (same) fixedAt = i;
    }
    if (stats.trackSweeps && cyclePeriod == 0) {
      unsigned long long hsh = 1469598103934665603ull;
      auto mix = [&](float f) { if (f == 0.0f) f = 0.0f; unsigned u; std::memcpy(&u, &f, 4); hsh = (hsh ^ u) * 1099511628211ull; };
      for (auto& vv : velocities) { mix(vv.v.x); mix(vv.v.y); mix(vv.w); }
      for (auto& cc : contactSolver.vcs) for (int q = 0; q < 2; ++q) { mix(cc.points[q].normalImpulse); mix(cc.points[q].tangentImpulse); }
      for (int jid : ijoints) { mix(joints[jid].impulse.x); mix(joints[jid].impulse.y); mix(joints[jid].impulse.z); mix(joints[jid].motorImpulse); }
      for (int pp = 1; pp <= 32 && pp <= (int)hashes.size(); ++pp)
        if (hashes[hashes.size() - pp] == hsh) { cyclePeriod = pp; cycleAt = i; break; }
      hashes.push_back(hsh);
    }
    if (stats.trackSweeps && traceOn()) {   // diagnostic (B2O_TRACE_NONE=1): the sweep states of solves that neither settle nor cycle
      std::vector<float> row;
      for (auto& vv : velocities) { row.push_back(vv.v.x); row.push_back(vv.v.y); row.push_back(vv.w); }
      for (auto& cc : contactSolver.vcs) for (int q = 0; q < cc.pointCount; ++q) { row.push_back(cc.points[q].normalImpulse); row.push_back(cc.points[q].tangentImpulse); }
      trace.push_back(row);
    }
  }
  if (stats.trackSweeps && traceOn() && !trace.empty()) {   // one line per solve: first sweep from which the velocities never change again
    const size_t nv = 3 * velocities.size();
    int vfix = (int)trace.size() - 1;
    while (vfix > 0 && std::equal(trace[vfix - 1].begin(), trace[vfix - 1].begin() + nv, trace.back().begin())) --vfix;
    std::fprintf(stderr, "VFIX %d %d %d %d %d %d %d\n", (int)ijoints.size(), (int)icontacts.size(), (int)ibodies.size(), fixedAt, cyclePeriod, cycleAt, vfix);
  }
  if (stats.trackSweeps && traceOn() && ijoints.empty() && fixedAt < 0 && cyclePeriod == 0 && traceLeft() > 0 &&
      !std::equal(trace[trace.size() - 2].begin(), trace[trace.size() - 2].begin() + 3 * velocities.size(), trace.back().begin())) {
    --traceLeft();
    std::fprintf(stderr, "TRACE nb=%d nc=%d\n", (int)ibodies.size(), (int)icontacts.size());
    for (size_t i = 0; i < trace.size(); ++i) {
      if (i >= 4 && i + 14 < trace.size() && i % 30 != 0) continue;
      std::fprintf(stderr, " %3d:", (int)i);
      for (float f : trace[i]) std::fprintf(stderr, " %a", f);
      std::fprintf(stderr, "\n");
    }
  }
  if (stats.trackSweeps) { stats.periodHist[cyclePeriod == 0 ? 33 : cyclePeriod]++; if (cyclePeriod) { stats.cycleAtSum += cycleAt; stats.cycleCount++; } }
  if (stats.trackSweeps) stats.sweepHist[fixedAt < 0 ? 181 : fixedAt + 1]++;
  if (stats.trackSweeps) {
    stats.solveLog.insert(stats.solveLog.end(), {fixedAt, cyclePeriod, cycleAt, (int)ijoints.size(), (int)icontacts.size(), (int)ibodies.size()});
    if (!ijoints.empty()) stats.nicHist[icontacts.size() < 15 ? icontacts.size() : 15]++;
    else stats.nicHistFree[icontacts.size() < 15 ? icontacts.size() : 15]++;
    int need = step.velocityIterations;
    if (fixedAt >= 0) need = fixedAt + 1;
    else if (ijoints.empty() && cyclePeriod >= 1 && cyclePeriod <= 4 && cycleAt < 24) need = cycleAt + 1;
    if (need > stats.lastSolveSweeps) stats.lastSolveSweeps = need;
  }
  contactSolver.StoreImpulses();
  for (int i = 0; i < bodyCount; ++i) {
    Vec2 c = positions[i].c;
    float a = positions[i].a;
    Vec2 v = velocities[i].v;
    float w = velocities[i].w;
    Vec2 translation = h * v;
    if (Dot(translation, translation) > kMaxTranslationSquared) {
      float ratio = kMaxTranslation / Length(translation);
      v *= ratio;
    }
    float rotation = h * w;
    if (rotation * rotation > kMaxRotationSquared) {
      float ratio = kMaxRotation / Abs(rotation);
      w *= ratio;
    }
    c += h * v;
    a += h * w;
    positions[i].c = c;
    positions[i].a = a;
    velocities[i].v = v;
    velocities[i].w = w;
  }
  bool positionSolved = false;
  int posItersUsed = 0;
  std::vector<std::vector<Position>> posHist;
  int posRepeatAt = -1;
  for (int i = 0; i < step.positionIterations; ++i) {
    ++posItersUsed;
    if (stats.trackSweeps && posRepeatAt < 0) {
      for (int pp = 1; pp <= 4 && pp <= (int)posHist.size(); ++pp) {
        const std::vector<Position>& old = posHist[posHist.size() - pp];
        bool same = true;
        for (int k = 0; k < bodyCount; ++k) same = same && old[k].c.x == positions[k].c.x && old[k].c.y == positions[k].c.y && old[k].a == positions[k].a;
        if (same) { posRepeatAt = i; break; }
      }
      posHist.push_back(positions);
    }
    bool contactsOkay = contactSolver.SolvePositionConstraints();
    bool jointsOkay = true;
    for (int jid : ijoints) {
      bool jointOkay = JointSolvePositionConstraints(joints[jid], positions.data());
      jointsOkay = jointsOkay && jointOkay;
    }
    if (contactsOkay && jointsOkay) {
      positionSolved = true;
      break;
    }
  }
  stats.posIterHist[positionSolved ? posItersUsed : 61]++;
  if (stats.trackSweeps && !positionSolved) stats.posFixHist[posRepeatAt < 0 ? 63 : posRepeatAt]++;
  for (int i = 0; i < bodyCount; ++i) {
    Body& body = bodies[ibodies[i]];
    body.sweep.c = positions[i].c;
    body.sweep.a = positions[i].a;
    body.v = velocities[i].v;
    body.w = velocities[i].w;
    body.SynchronizeTransform();
  }
  // allowSleep
  {
    float minSleepTime = kMaxFloat;
    const float linTolSqr = kLinearSleepTolerance * kLinearSleepTolerance;
    const float angTolSqr = kAngularSleepTolerance * kAngularSleepTolerance;
    for (int i = 0; i < bodyCount; ++i) {
      Body& b = bodies[ibodies[i]];
      if (b.type == kStaticBody) continue;
      if (b.w * b.w > angTolSqr || Dot(b.v, b.v) > linTolSqr) {
        b.sleepTime = 0.0f;
        minSleepTime = 0.0f;
      } else {
        b.sleepTime += h;
        minSleepTime = Min(minSleepTime, b.sleepTime);
      }
    }
    if (minSleepTime >= kTimeToSleep && positionSolved) {
      for (int i = 0; i < bodyCount; ++i) bodies[ibodies[i]].SetAwake(false);
    }
  }
}

// b2World::Solve
inline void World::Solve(const TimeStep& step) {
  for (Body& b : bodies) b.islandFlag = false;
  for (int cid : contactList) contacts[cid].islandFlag = false;
  for (Joint& j : joints) j.islandFlag = false;
  std::vector<int> stack;
  std::vector<int> ibodies, icontacts, ijoints;
  for (int seed = (int)bodies.size() - 1; seed >= 0; --seed) {  // m_bodyList: last created first
    Body& sb = bodies[seed];
    if (sb.islandFlag) continue;
    if (!sb.awake) continue;
    if (sb.type == kStaticBody) continue;
    ibodies.clear();
    icontacts.clear();
    ijoints.clear();
    stack.clear();
    stack.push_back(seed);
    sb.islandFlag = true;
    while (!stack.empty()) {
      int bi = stack.back();
      stack.pop_back();
      Body& b = bodies[bi];
      b.islandIndex = (int)ibodies.size();
      ibodies.push_back(bi);
      b.SetAwake(true);
      if (b.type == kStaticBody) continue;
      for (int cid : b.contacts) {
        Contact& c = contacts[cid];
        if (c.islandFlag) continue;
        if (!c.enabled || !c.touching) continue;
        icontacts.push_back(cid);
        c.islandFlag = true;
        int other = c.bodyA == bi ? c.bodyB : c.bodyA;
        if (bodies[other].islandFlag) continue;
        stack.push_back(other);
        bodies[other].islandFlag = true;
      }
      for (int jid : b.joints) {
        Joint& j = joints[jid];
        if (j.islandFlag) continue;
        int other = j.bodyA == bi ? j.bodyB : j.bodyA;
        ijoints.push_back(jid);
        j.islandFlag = true;
        if (bodies[other].islandFlag) continue;
        stack.push_back(other);
        bodies[other].islandFlag = true;
      }
    }
    IslandSolve(ibodies, icontacts, ijoints, step);
    stats.islands++;
    for (int bi : ibodies)
      if (bodies[bi].type == kStaticBody) bodies[bi].islandFlag = false;
  }
  for (int bi = (int)bodies.size() - 1; bi >= 0; --bi) {
    Body& b = bodies[bi];
    if (!b.islandFlag) continue;
    if (b.type == kStaticBody) continue;
    SynchronizeFixtures(bi);
  }
  FindNewContacts();
}

// b2Island::SolveTOI
inline void World::IslandSolveTOI(std::vector<int>& ibodies, std::vector<int>& icontacts, const TimeStep& subStep, int toiIndexA,
                                  int toiIndexB) {
  int bodyCount = (int)ibodies.size();
  std::vector<Position> positions(bodyCount);
  std::vector<Velocity> velocities(bodyCount);
  for (int i = 0; i < bodyCount; ++i) {
    Body& b = bodies[ibodies[i]];
    positions[i].c = b.sweep.c;
    positions[i].a = b.sweep.a;
    velocities[i].v = b.v;
    velocities[i].w = b.w;
  }
  std::vector<Contact*> cs;
  for (int cid : icontacts) cs.push_back(&contacts[cid]);
  ContactSolver contactSolver;
  contactSolver.Init(subStep, cs, this, positions.data(), velocities.data());
  for (int i = 0; i < subStep.positionIterations; ++i) {
    bool contactsOkay = contactSolver.SolveTOIPositionConstraints(toiIndexA, toiIndexB);
    if (contactsOkay) break;
  }
  bodies[ibodies[toiIndexA]].sweep.c0 = positions[toiIndexA].c;
  bodies[ibodies[toiIndexA]].sweep.a0 = positions[toiIndexA].a;
  bodies[ibodies[toiIndexB]].sweep.c0 = positions[toiIndexB].c;
  bodies[ibodies[toiIndexB]].sweep.a0 = positions[toiIndexB].a;
  contactSolver.InitializeVelocityConstraints();
  for (int i = 0; i < subStep.velocityIterations; ++i) contactSolver.SolveVelocityConstraints();
  float h = subStep.dt;
  for (int i = 0; i < bodyCount; ++i) {
    Vec2 c = positions[i].c;
    float a = positions[i].a;
    Vec2 v = velocities[i].v;
    float w = velocities[i].w;
    Vec2 translation = h * v;
    if (Dot(translation, translation) > kMaxTranslationSquared) {
      float ratio = kMaxTranslation / Length(translation);
      v *= ratio;
    }
    float rotation = h * w;
    if (rotation * rotation > kMaxRotationSquared) {
      float ratio = kMaxRotation / Abs(rotation);
      w *= ratio;
    }
    c += h * v;
    a += h * w;
    positions[i].c = c;
    positions[i].a = a;
    velocities[i].v = v;
    velocities[i].w = w;
    Body& body = bodies[ibodies[i]];
    body.sweep.c = c;
    body.sweep.a = a;
    body.v = v;
    body.w = w;
    body.SynchronizeTransform();
  }
}

// b2World::SolveTOI
inline void World::SolveTOI(const TimeStep& step) {
  for (Body& b : bodies) {
    b.islandFlag = false;
    b.sweep.alpha0 = 0.0f;
  }
  for (int cid : contactList) {
    Contact& c = contacts[cid];
    c.toiFlag = false;
    c.islandFlag = false;
    c.toiCount = 0;
    c.toi = 1.0f;
  }
  for (;;) {
    int minContact = -1;
    float minAlpha = 1.0f;
    for (int cid : contactList) {
      Contact& c = contacts[cid];
      if (!c.enabled) continue;
      if (c.toiCount > kMaxSubSteps) continue;
      float alpha = 1.0f;
      if (c.toiFlag) {
        alpha = c.toi;
      } else {
        Body& bA = bodies[c.bodyA];
        Body& bB = bodies[c.bodyB];
        int typeA = bA.type, typeB = bB.type;
        bool activeA = bA.awake && typeA != kStaticBody;
        bool activeB = bB.awake && typeB != kStaticBody;
        if (!activeA && !activeB) continue;
        bool collideA = typeA != kDynamicBody;  // no bullets in boxLCD
        bool collideB = typeB != kDynamicBody;
        if (!collideA && !collideB) continue;
        float alpha0 = bA.sweep.alpha0;
        if (bA.sweep.alpha0 < bB.sweep.alpha0) {
          alpha0 = bB.sweep.alpha0;
          bA.sweep.Advance(alpha0);
        } else if (bB.sweep.alpha0 < bA.sweep.alpha0) {
          alpha0 = bA.sweep.alpha0;
          bB.sweep.Advance(alpha0);
        }
        DistanceProxy proxyA, proxyB;
        proxyA.Set(&bA.shape);
        proxyB.Set(&bB.shape);
        TOIOutput output;
        TimeOfImpact(&output, &proxyA, bA.sweep, &proxyB, bB.sweep, 1.0f);
        stats.toiCalls++;
        float beta = output.t;
        if (output.state == kTOITouching) {
          alpha = Min(alpha0 + (1.0f - alpha0) * beta, 1.0f);
        } else {
          alpha = 1.0f;
        }
        c.toi = alpha;
        c.toiFlag = true;
      }
      if (alpha < minAlpha) {
        minContact = cid;
        minAlpha = alpha;
      }
    }
    if (minContact < 0 || 1.0f - 10.0f * kEpsilon < minAlpha) break;

    Contact& mc = contacts[minContact];
    int ia = mc.bodyA, ib = mc.bodyB;
    Body& bA = bodies[ia];
    Body& bB = bodies[ib];
    Sweep backup1 = bA.sweep;
    Sweep backup2 = bB.sweep;
    stats.toiIters++;
    bA.Advance(minAlpha);
    bB.Advance(minAlpha);
    UpdateContact(minContact);
    mc.toiFlag = false;
    ++mc.toiCount;
    if (!mc.enabled || !mc.touching) {
      mc.enabled = false;
      bA.sweep = backup1;
      bB.sweep = backup2;
      bA.SynchronizeTransform();
      bB.SynchronizeTransform();
      continue;
    }
    bA.SetAwake(true);
    bB.SetAwake(true);
    stats.toiEvents++;
    std::vector<int> ibodies, icontacts;
    bA.islandIndex = 0;
    ibodies.push_back(ia);
    bB.islandIndex = 1;
    ibodies.push_back(ib);
    icontacts.push_back(minContact);
    bA.islandFlag = true;
    bB.islandFlag = true;
    mc.islandFlag = true;
    int pair[2] = {ia, ib};
    for (int k = 0; k < 2; ++k) {
      int bi = pair[k];
      Body& body = bodies[bi];
      if (body.type == kDynamicBody) {
        for (int cid : body.contacts) {
          if ((int)ibodies.size() == 2 * kMaxTOIContacts) break;
          if ((int)icontacts.size() == kMaxTOIContacts) break;
          Contact& contact = contacts[cid];
          if (contact.islandFlag) continue;
          int oi = contact.bodyA == bi ? contact.bodyB : contact.bodyA;
          Body& other = bodies[oi];
          if (other.type == kDynamicBody) continue;  // no bullets
          Sweep backup = other.sweep;
          if (!other.islandFlag) other.Advance(minAlpha);
          UpdateContact(cid);
          if (!contact.enabled) {
            other.sweep = backup;
            other.SynchronizeTransform();
            continue;
          }
          if (!contact.touching) {
            other.sweep = backup;
            other.SynchronizeTransform();
            continue;
          }
          contact.islandFlag = true;
          icontacts.push_back(cid);
          if (other.islandFlag) continue;
          other.islandFlag = true;
          if (other.type != kStaticBody) other.SetAwake(true);
          other.islandIndex = (int)ibodies.size();
          ibodies.push_back(oi);
        }
      }
    }
    TimeStep subStep;
    subStep.dt = (1.0f - minAlpha) * step.dt;
    subStep.inv_dt = 1.0f / subStep.dt;
    subStep.dtRatio = 1.0f;
    subStep.positionIterations = 20;
    subStep.velocityIterations = step.velocityIterations;
    subStep.warmStarting = false;
    IslandSolveTOI(ibodies, icontacts, subStep, bA.islandIndex, bB.islandIndex);
    for (int bi : ibodies) {
      Body& body = bodies[bi];
      body.islandFlag = false;
      if (body.type != kDynamicBody) continue;
      SynchronizeFixtures(bi);
      for (int cid : body.contacts) {
        contacts[cid].toiFlag = false;
        contacts[cid].islandFlag = false;
      }
    }
    FindNewContacts();
  }
}

}  // namespace b2o
