// Exercises include/AreaAverageInterpolation.hpp exactly the way the reference's main() drives its class
// (Source.cpp:1558-1570).  Modes:
//   shim_test errors            argument-error paths: {false, message}, dst / dstIsocenter untouched
//   shim_test run <mode> <W> <H> <srcRes> <dstRes> <isoX> <isoY> <angle> <seed>
//                               resample the Appendix-C.1 synthetic image and print "dW dH isoX isoY" then the
//                               output values with 17 significant digits (compared with the oracle by pytest)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "AreaAverageInterpolation.hpp"

static IMG synth(int W, int H, uint64_t seed)
{
    IMG img(H, std::vector<double>(W));
    const uint64_t G = 0x9E3779B97F4A7C15ull;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint64_t z = seed * G + ((uint64_t)y * W + x) + G;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            img[y][x] = (double)((float)(z >> 40) * 0x1p-24f);
        }
    return img;
}

static int check(bool cond, const char *what)
{
    if (!cond) { std::printf("FAIL: %s\n", what); return 1; }
    return 0;
}

int main(int argc, char **argv)
{
    AreaAverageInterpolation aa;
    if (argc >= 2 && !std::strcmp(argv[1], "errors")) {
        int bad = 0;
        IMG src = synth(6, 5, 1), dst(2, std::vector<double>(3, -7.0));
        dP iso(-1.5, -2.5);
        auto untouched = [&]() { return dst.size() == 2 && dst[0].size() == 3 && dst[1][2] == -7.0 && iso.first == -1.5 && iso.second == -2.5; };
        auto r = aa.areaAverageInterpolation(src, dst, {1, 2}, {1, 1}, {0, 0}, iso, 0);
        bad += check(!r.first && r.second == "Assumed X & Y resolution are same." && untouched(), "x/y resolution mismatch (Source.cpp:112-117)");
        r = aa.fastAreaAverageInterpolation(src, dst, {1, 1}, {2, 2.5}, {0, 0}, iso, 0);
        bad += check(!r.first && r.second == "Assumed X & Y resolution are same." && untouched(), "dst resolution mismatch");
        r = aa.areaAverageInterpolation(src, dst, {0, 0}, {1, 1}, {0, 0}, iso, 0);
        bad += check(!r.first && r.second == "0 or negative resolution is not acceptable." && untouched(), "zero resolution (Source.cpp:118-122)");
        r = aa.areaAverageInterpolation(src, dst, {1, 1}, {-3, -3}, {0, 0}, iso, 0);
        bad += check(!r.first && r.second == "0 or negative resolution is not acceptable." && untouched(), "negative resolution");
        r = aa.areaAverageInterpolation(IMG(), dst, {1, 1}, {1, 1}, {0, 0}, iso, 0);
        bad += check(!r.first && r.second == "There is no data in src array." && untouched(), "no rows (Source.cpp:123-127)");
        r = aa.fastAreaAverageInterpolation(IMG(3), dst, {1, 1}, {1, 1}, {0, 0}, iso, 0);
        bad += check(!r.first && r.second == "There is no data in the second dimension of src array." && untouched(), "empty first row (Source.cpp:128-132)");
        // the resolution checks come before the emptiness checks, like in the reference
        r = aa.areaAverageInterpolation(IMG(), dst, {1, 2}, {1, 1}, {0, 0}, iso, 0);
        bad += check(!r.first && r.second == "Assumed X & Y resolution are same.", "check order");
        IMG ragged = src;
        ragged[2].pop_back();
        r = aa.areaAverageInterpolation(ragged, dst, {1, 1}, {1, 1}, {0, 0}, iso, 0);
        bad += check(!r.first && r.second == "Ragged src array." && untouched(), "ragged rows are rejected");
        r = aa.areaAverageInterpolation(src, dst, {1, 1}, {1, 1}, {0.0 / 0.0, 0}, iso, 0);
        bad += check(!r.first && r.second == "Non-finite argument." && untouched(), "NaN isocenter is rejected");
        std::printf(bad ? "errors: %d failed\n" : "errors: all ok\n", bad);
        return bad ? 1 : 0;
    }
    if (argc == 11 && !std::strcmp(argv[1], "run")) {
        const int mode = std::atoi(argv[2]), W = std::atoi(argv[3]), H = std::atoi(argv[4]);
        const double sr = std::atof(argv[5]), dr = std::atof(argv[6]), ix = std::atof(argv[7]), iy = std::atof(argv[8]), ang = std::atof(argv[9]);
        IMG src = synth(W, H, (uint64_t)std::atoll(argv[10])), dst;
        dP iso(0, 0);
        auto r = mode == 2 ? aa.fastAreaAverageInterpolation(src, dst, {sr, sr}, {dr, dr}, {ix, iy}, iso, ang)
                           : aa.areaAverageInterpolation(src, dst, {sr, sr}, {dr, dr}, {ix, iy}, iso, ang);
        if (!r.first) { std::printf("ERROR %s\n", r.second.c_str()); return 2; }
        std::printf("%zu %zu %.17g %.17g\n", dst.empty() ? (size_t)0 : dst[0].size(), dst.size(), iso.first, iso.second);
        for (const auto &row : dst) {
            for (double v : row) std::printf("%.17g ", v);
            std::printf("\n");
        }
        return 0;
    }
    std::printf("usage: shim_test errors | run <mode> <W> <H> <srcRes> <dstRes> <isoX> <isoY> <angle> <seed>\n");
    return 64;
}
