// ref_glm_dump -- evaluates, with the reference's own vendored glm (compiled
// from /root/reference/src/glm where it lies), the glm expressions that
// Camera::update (camera.cpp:89-95) and Scene::Light's constructor
// (scene.h:32-34) consist of, and prints the resulting f32 bit patterns.
// camera.cpp / scene.h themselves cannot be compiled here (they include
// GL/glew.h, absent from the image), so this pins the glm arithmetic they
// delegate to, not the files.
//
// TEST INFRASTRUCTURE ONLY.
//   usage: ref_glm_dump basis ex ey ez fx fy fz
//          ref_glm_dump light px py pz rx ry rz sx sy sz
//          ref_glm_dump length x y z | acos x | sin x | cos x | atan x
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtx/euler_angles.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>

static unsigned bits(float f)
{
    unsigned u;
    std::memcpy(&u, &f, 4);
    return u;
}
static float arg(char **argv, int i) { return (float)std::atof(argv[i]); }
static void show(const char *tag, const glm::vec3 &v)
{
    std::printf("%s %08x %08x %08x\n", tag, bits(v.x), bits(v.y), bits(v.z));
}

int main(int argc, char **argv)
{
    const char *op = argc > 1 ? argv[1] : "";
    if (argc == 8 && !std::strcmp(op, "basis")) {
        glm::vec3 eye(arg(argv, 2), arg(argv, 3), arg(argv, 4));
        glm::vec3 focus(arg(argv, 5), arg(argv, 6), arg(argv, 7));
        glm::vec3 fwd = glm::normalize(focus - eye);
        glm::vec3 rgt = glm::normalize(glm::vec3(-fwd.z, 0.f, fwd.x));
        glm::vec3 upv = glm::normalize(glm::cross(rgt, fwd));
        show("forward", fwd);
        show("up", upv);
        show("right", rgt);
        return 0;
    }
    if (argc == 11 && !std::strcmp(op, "light")) {
        glm::vec3 pos(arg(argv, 2), arg(argv, 3), arg(argv, 4));
        glm::vec3 rot(arg(argv, 5), arg(argv, 6), arg(argv, 7));
        glm::vec3 scl(arg(argv, 8), arg(argv, 9), arg(argv, 10));
        glm::mat4 m = glm::translate(glm::mat4(1.f), pos) * glm::eulerAngleYXZ(rot.y, rot.x, rot.z) *
                      glm::scale(glm::mat4(1.f), scl);
        for (int c = 0; c < 4; ++c)
            std::printf("col%d %08x %08x %08x %08x\n", c, bits(m[c][0]), bits(m[c][1]), bits(m[c][2]),
                        bits(m[c][3]));
        return 0;
    }
    if (argc == 5 && !std::strcmp(op, "length")) {
        std::printf("%08x\n", bits(glm::length(glm::vec3(arg(argv, 2), arg(argv, 3), arg(argv, 4)))));
        return 0;
    }
    if (argc == 3) {
        float x = arg(argv, 2), r;
        if (!std::strcmp(op, "acos")) r = glm::acos(x);
        else if (!std::strcmp(op, "sin")) r = glm::sin(x);
        else if (!std::strcmp(op, "cos")) r = glm::cos(x);
        else if (!std::strcmp(op, "atan")) r = glm::atan(x);
        else return 2;
        std::printf("%08x\n", bits(r));
        return 0;
    }
    std::fprintf(stderr, "usage: %s basis e f | light p r s | length v | acos|sin|cos|atan x\n", argv[0]);
    return 2;
}
