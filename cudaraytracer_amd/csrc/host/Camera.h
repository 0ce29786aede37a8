// Camera of the host API: position + right/up/look basis and the view matrix
// whose inverse goes to the kernel (cudaTracer/Camera.cpp:7-147,
// cudaTracer/RayTracer.cpp:201-202).  The GUI hooks of the reference's
// constructor are gone; everything else behaves the same.
#ifndef RT_HOST_CAMERA_H
#define RT_HOST_CAMERA_H

#include "Mat.h"
#include "Vec.h"

namespace rt {

class Camera {
   public:
    Camera() { calcView(); }

    // Re-orthonormalises the basis around the look direction, then rebuilds the view matrix.
    void update() {
        vecLookAt = vecLookAt.normalized();
        vecUp = vecLookAt.cross(vecRight).normalized();
        vecRight = vecUp.cross(vecLookAt).normalized();
        calcView();
    }

    Mat44f getCamera() const { return matView; }
    Vec3f getPosition() const { return vecPosition; }
    Vec3f getRight() const { return vecRight; }
    void getXZ(float* x, float* z) const {
        *x = vecPosition[0];
        *z = vecPosition[2];
    }

    void setPosition(float x, float y, float z) { vecPosition = Vec3f(x, y, z); }
    void setPosition(Vec3f pos) { vecPosition = pos; }
    void setZ(float z) { vecPosition[2] = z; }
    void setLookAt(Vec3f target) { setDirection(target - vecPosition); }
    void setDirection(Vec3f direction) {
        vecUp = Vec3f(0.0f, 1.0f, 0.0f);
        vecLookAt = direction;
        vecRight = vecUp.cross(vecLookAt);
        vecUp = vecLookAt.cross(vecRight);
        vecRight = vecRight.normalized();
        vecUp = vecUp.normalized();
        vecLookAt = vecLookAt.normalized();
        update();
    }

    void strafe(float amount) { vecPosition += amount * vecRight; }
    void walk(float amount) { vecPosition += amount * vecLookAt; }
    void ascend(float amount) { vecPosition += amount * vecUp; }
    void pitch(float angle) {
        const Mat44f rot = Mat44f::rotation(vecRight, angle);
        vecUp = rot.multVec(vecUp);
        vecLookAt = rot.multVec(vecLookAt);
    }
    void rotateY(float angle) {
        const Mat44f rot = Mat44f::rotationY(angle);
        vecRight = rot.multVec(vecRight);
        vecUp = rot.multVec(vecUp);
        vecLookAt = rot.multVec(vecLookAt);
    }

   private:
    Vec3f vecPosition = Vec3f(0.0f, 0.0f, 0.0f);
    Vec3f vecUp = Vec3f(0.0f, 1.0f, 0.0f);
    Vec3f vecLookAt = Vec3f(0.0f, 0.0f, 1.0f);
    Vec3f vecRight = Vec3f(1.0f, 0.0f, 0.0f);
    Mat44f matView;

    // Basis vectors go into the COLUMNS of rows 0-2, the position into row 3
    // (cudaTracer/Camera.cpp:69-92).
    void calcView() {
        for (int r = 0; r < 3; ++r) {
            matView[r][0] = vecRight[r];
            matView[r][1] = vecUp[r];
            matView[r][2] = vecLookAt[r];
            matView[r][3] = 0.0f;
            matView[3][r] = vecPosition[r];
        }
        matView[3][3] = 1.0f;
    }
};

}  // namespace rt
#endif
