// AreaAverageInterpolation.hpp -- header-only C++ drop-in for the reference class, on top of the C ABI
// (include/aai.h, libaai_hip.so).
//
// The reference's public surface is one stateless class with two methods of identical signature
// (Source.cpp:52-57 and 584-586), driven from its main() at Source.cpp:1558-1570:
//
//     AreaAverageInterpolation aa;
//     ret = aa.areaAverageInterpolation    (src, dst, srcResolution, dstResolution, srcIsocenter, dstIsocenter, rotationAngle);
//     ret = aa.fastAreaAverageInterpolation(src, dst, srcResolution, dstResolution, srcIsocenter, dstIsocenter, rotationAngle);
//
// This header re-creates exactly that: same type aliases (Source.cpp:30-50), same method names, argument
// order and meaning, same return convention ({true,""} / {false,message} with the reference's four
// messages, Source.cpp:115,120,125,130), `dst` cleared and resized by the callee (Source.cpp:411-414),
// `dstIsocenter` overwritten with integer-valued doubles (Source.cpp:185-186), and both left untouched on
// failure.  A caller written against the reference compiles unchanged after replacing the class
// definition by `#include "AreaAverageInterpolation.hpp"` and linking -laai_hip.
//
// Differences, all documented in DESIGN.md:
//   * the work runs on the GPU through aai_resample_f64 (fp32 pixels, fp64 geometry): results agree with
//     the reference to 1e-5 relative, not bit for bit;
//   * the parameter banner the reference prints to cout on every call (Source.cpp:59-75) is printed only
//     when `verbose` is set;
//   * conditions the reference cannot report (non-finite arguments, no GPU, HIP failure) come back as
//     {false, message} too instead of undefined behaviour;
//   * ragged `src` rows (undefined behaviour in the reference, which sizes by src.front(), Source.cpp:150)
//     are rejected with {false, "Ragged src array."}.
#pragma once

#include <iomanip>
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "aai.h"

/* Alias for 2D-image (row-major [y][x]), as Source.cpp:31 */
using IMG = std::vector<std::vector<double>>;
/* Paired data, first: x, second: y, as Source.cpp:36-46 */
using uiP = std::pair<unsigned int, unsigned int>;
using iP = std::pair<int, int>;
using dP = std::pair<double, double>;

class AreaAverageInterpolation {
public:
    bool verbose = false;                 // print the reference's parameter banner
    int policy = AAI_POLICY_REFERENCE;    // AAI_POLICY_EXACT for geometrically exact areas; | AAI_POLICY_DOUBLE_PRECISION: see aai.h

    std::pair<bool, std::string> areaAverageInterpolation(IMG src, IMG &dst, dP srcResolution, dP dstResolution,
                                                          dP srcIsocenter, dP &dstIsocenter, double rotationAngle)
    {
        return run(AAI_MODE_AREA, "areaAverageInterpolation    ", src, dst, srcResolution, dstResolution, srcIsocenter,
                   dstIsocenter, rotationAngle);
    }

    std::pair<bool, std::string> fastAreaAverageInterpolation(IMG src, IMG &dst, dP srcResolution, dP dstResolution,
                                                              dP srcIsocenter, dP &dstIsocenter, double rotationAngle)
    {
        return run(AAI_MODE_FAST, "fastAreaAverageInterpolation", src, dst, srcResolution, dstResolution, srcIsocenter,
                   dstIsocenter, rotationAngle);
    }

    // Build-defined comparison paths (the reference names them in README.md:8 but implements neither).
    std::pair<bool, std::string> bilinearInterpolation(IMG src, IMG &dst, dP srcResolution, dP dstResolution,
                                                       dP srcIsocenter, dP &dstIsocenter, double rotationAngle)
    {
        return run(AAI_MODE_BILINEAR, "bilinearInterpolation       ", src, dst, srcResolution, dstResolution, srcIsocenter,
                   dstIsocenter, rotationAngle);
    }
    std::pair<bool, std::string> bicubicInterpolation(IMG src, IMG &dst, dP srcResolution, dP dstResolution,
                                                      dP srcIsocenter, dP &dstIsocenter, double rotationAngle)
    {
        return run(AAI_MODE_BICUBIC, "bicubicInterpolation        ", src, dst, srcResolution, dstResolution, srcIsocenter,
                   dstIsocenter, rotationAngle);
    }

private:
    std::pair<bool, std::string> run(int mode, const char *name, const IMG &src, IMG &dst, dP srcResolution,
                                     dP dstResolution, dP srcIsocenter, dP &dstIsocenter, double rotationAngle) const
    {
        if (verbose) banner(name, srcResolution, dstResolution, srcIsocenter, rotationAngle);

        aai_request rq{};
        rq.mode = mode;
        rq.policy = policy;
        rq.src_height = (int32_t)src.size();
        rq.src_width = src.empty() ? 0 : (int32_t)src.front().size();
        rq.src_res_x = srcResolution.first;  rq.src_res_y = srcResolution.second;
        rq.dst_res_x = dstResolution.first;  rq.dst_res_y = dstResolution.second;
        rq.src_iso_x = srcIsocenter.first;   rq.src_iso_y = srcIsocenter.second;
        rq.rotation_deg = rotationAngle;

        aai_layout lay{};
        if (aai_query(&rq, &lay) != AAI_OK) return {false, aai_last_error()};
        for (const auto &row : src)
            if (row.size() != (size_t)rq.src_width) return {false, "Ragged src array."};

        // IMG is a vector of separately allocated rows: flatten for the ABI, un-flatten the result.
        std::vector<double> flat((size_t)rq.src_width * rq.src_height);
        for (int y = 0; y < rq.src_height; ++y)
            std::copy(src[y].begin(), src[y].end(), flat.begin() + (size_t)y * rq.src_width);
        std::vector<double> out((size_t)lay.dst_width * lay.dst_height);
        if (aai_resample_f64(&rq, flat.data(), rq.src_width, out.data(), lay.dst_width > 0 ? lay.dst_width : 1, &lay) != AAI_OK)
            return {false, aai_last_error()};

        dst.clear();
        dst.resize((size_t)lay.dst_height);
        for (int y = 0; y < lay.dst_height; ++y)
            dst[y].assign(out.begin() + (size_t)y * lay.dst_width, out.begin() + (size_t)(y + 1) * lay.dst_width);
        dstIsocenter = std::make_pair(lay.dst_iso_x, lay.dst_iso_y);
        return {true, ""};
    }

    static void banner(const char *name, dP sr, dP dr, dP iso, double angle)
    {
        using std::cout; using std::endl; using std::setw;
        cout << "**********************************************************" << endl;
        cout << "* AreaAverageInterpolation::" << name << " *" << endl;
        cout << std::setprecision(10);
        cout << "* Input parameters                                       *" << endl;
        cout << "*                                                        *" << endl;
        cout << "* srcResolution : " << setw(9) << sr.first << ", " << setw(9) << sr.second << setw(20) << " [pixel/mm or dpi] *" << endl;
        cout << "* dstResolution : " << setw(9) << dr.first << ", " << setw(9) << dr.second << setw(20) << " [pixel/mm or dpi] *" << endl;
        cout << "* srcIsocenter  : " << setw(9) << iso.first << ", " << setw(9) << iso.second << setw(20) << " [pixels] *" << endl;
        cout << "* rotationAngle : " << setw(20) << angle << setw(20) << " [degrees] *" << endl;
        cout << "**********************************************************" << endl;
    }
};
