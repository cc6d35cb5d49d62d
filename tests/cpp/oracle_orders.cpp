// tests/cpp/oracle_orders.cpp — the evaluation orders that oracle/PINNING.md read off the reference's prebuilt binary, held in place:
// inputs on which the candidate associations of a sum round differently, and the result the binary's order gives. (The quaternion
// product's grouping and SO3::exp's branch test move results by less than any input can expose on purpose: they are held by the
// golden vectors, tests/golden/icp_small.npz.)
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "locref_math.hpp"

using namespace locref;

#define CHECK(c, msg)                                                       \
    do {                                                                    \
        if (!(c)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, msg); return 1; } \
    } while (0)

int main() {
    // Vector3d reductions: (x + y) + z — with x = 1e16, y = z = 1: (1e16 + 1) + 1 = 1e16, while 1e16 + (1 + 1) = 1e16 + 2
    {
        const V3 a{1e16, 1.0, 1.0}, ones{1.0, 1.0, 1.0};
        volatile double want = 1e16;
        want = want + 1.0;
        want = want + 1.0;
        CHECK(dot(a, ones) == want && dot(a, ones) != 1e16 + 2.0, "dot(): not (x + y) + z");
    }
    // dx.norm(): (d0² + (d2² + d4²)) + (d1² + (d3² + d5²)) — d0² = 1e16, d2² = d4² = 1: the inner pair first gives 1e16 + 2
    {
        const double d[6] = {1e8, 0.0, 1.0, 0.0, 1.0, 0.0};
        CHECK(norm6(d) == std::sqrt(1e16 + 2.0), "norm6(): not (d0² + (d2² + d4²)) + (d1² + (d3² + d5²))");
        const double left_to_right = std::sqrt(((1e16 + 1.0) + 1.0));
        CHECK(norm6(d) != left_to_right, "norm6(): a left-to-right sum would have given the other value");
    }
    printf("oracle orders ok\n");
    return 0;
}
