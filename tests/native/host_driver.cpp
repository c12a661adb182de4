// Sanitizer driver for the host-side entry points of libsmrt_hip.so (host_assemble.cpp): built by
// tests/test_host_native_sanitized.py with g++ -fsanitize=address,undefined and run on the CPU.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <vector>

#include "../../include/smrt_hip.h"

#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
    const int64_t n = 50021, m = 4099, d = 7;
    std::vector<float> X((size_t)n * d), S((size_t)m * d);
    std::vector<double> Sd((size_t)m * d);
    for (size_t i = 0; i < X.size(); ++i) X[i] = (float)(i % 1013) * 0.25f;
    for (size_t i = 0; i < S.size(); ++i) { S[i] = -(float)(i % 509); Sd[i] = S[i]; }
    std::vector<int64_t> order((size_t)(n + m));
    for (int64_t i = 0; i < n + m; ++i) order[(size_t)i] = (i * 7919) % (n + m);            // a permutation: 7919 is coprime
    auto src = [&](int64_t r, int64_t j) { return r < n ? (double)X[(size_t)(r * d + j)] : (double)S[(size_t)((r - n) * d + j)]; };

    std::vector<double> out((size_t)(n + m) * d, -1.0), out2(out.size(), -1.0), out3(out.size(), -1.0);
    CHECK(smrt_assemble_rows(X.data(), 1, n, d, d, S.data(), m, d, order.data(), n + m, out.data(), 8) == 0);
    for (int64_t i = 0; i < n + m; ++i) for (int64_t j = 0; j < d; ++j) CHECK(out[(size_t)(i * d + j)] == src(order[(size_t)i], j));
    CHECK(smrt_assemble_rows_part(X.data(), 1, n, d, d, nullptr, m, d, order.data(), n + m, out2.data(), 5, 1) == 0);
    for (int64_t i = 0; i < n + m; ++i) CHECK(out2[(size_t)(i * d)] == (order[(size_t)i] < n ? src(order[(size_t)i], 0) : -1.0));
    CHECK(smrt_assemble_rows_part(nullptr, 1, n, d, d, S.data(), m, d, order.data(), n + m, out2.data(), 5, 2) == 0);
    CHECK(out2 == out);
    CHECK(smrt_assemble_rows_f64(X.data(), 1, n, d, d, Sd.data(), m, d, nullptr, n + m, out3.data(), 3) == 0);
    for (int64_t i = 0; i < n + m; ++i) CHECK(out3[(size_t)(i * d + 3)] == src(i, 3));
    // rejected arguments: nothing is read or written
    order[5] = n + m;
    CHECK(smrt_assemble_rows(X.data(), 1, n, d, d, S.data(), m, d, order.data(), n + m, out.data(), 8) == -1);
    order[5] = -1;
    CHECK(smrt_assemble_rows(X.data(), 1, n, d, d, S.data(), m, d, order.data(), n + m, out.data(), 8) == -1);
    CHECK(smrt_assemble_rows(X.data(), 1, n, d, d - 1, S.data(), m, d, nullptr, n + m, out.data(), 8) == -1);
    CHECK(smrt_assemble_rows(X.data(), 1, n, d, d, S.data(), m, d, nullptr, n, out.data(), 8) == -1);
    CHECK(smrt_assemble_rows_part(X.data(), 1, n, d, d, S.data(), m, d, nullptr, n + m, out.data(), 8, 3) == -1);

    // finiteness sweep: every position of a range that spans several threads, both dtypes
    std::vector<float> F((size_t)1 << 21, 1.5f);
    std::vector<double> D((size_t)1 << 20, -2.5);
    F[0] = std::numeric_limits<float>::max(); D[0] = -std::numeric_limits<double>::max();
    CHECK(smrt_all_finite(F.data(), 1, (int64_t)F.size(), 8) == 1);
    CHECK(smrt_all_finite(D.data(), 0, (int64_t)D.size(), 8) == 1);
    const size_t pos[] = {0, 1, 262143, 262144, F.size() / 2 + 3, F.size() - 1};
    for (size_t p : pos) {
        F[p] = NAN; CHECK(smrt_all_finite(F.data(), 1, (int64_t)F.size(), 8) == 0);
        F[p] = -INFINITY; CHECK(smrt_all_finite(F.data(), 1, (int64_t)F.size(), 3) == 0);
        F[p] = 1.0f;
        const size_t q = p % D.size();
        D[q] = NAN; CHECK(smrt_all_finite(D.data(), 0, (int64_t)D.size(), 8) == 0);
        D[q] = INFINITY; CHECK(smrt_all_finite(D.data(), 0, (int64_t)D.size(), 1) == 0);
        D[q] = 0.0;
    }
    CHECK(smrt_all_finite(F.data(), 1, 0, 4) == 1);
    CHECK(smrt_all_finite(nullptr, 1, 4, 4) == -1);
    std::printf("host driver ok\n");
    return 0;
}
