// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or executed from the product path.
//
// Harness that drives the UNMODIFIED reference implementation
// (/root/reference/Source.cpp, class AreaAverageInterpolation, Source.cpp:52-1432)
// through a small C ABI so that
//   * tests/golden/make_golden.py can generate golden vectors, and
//   * tests can pin oracle/aai_oracle.c against the real reference when it is present.
//
// The reference is compiled from where it lies (REF_SOURCE is passed as an absolute path by
// oracle/Makefile); no reference source text is copied into this repository.  The build output
// goes to oracle/_ref/ (git-ignored).  /root/reference does not exist on the GPU box: only the
// prebuilt oracle/_ref/libaai_ref.so travels there.
//
// The reference's own `main` (Source.cpp:1434) is renamed out of the way by macro; its banner
// printing (Source.cpp:59-75, 588-604) is silenced by swapping cout's streambuf during the call.

#ifndef REF_SOURCE
#error "REF_SOURCE must be defined to the absolute path of the reference Source.cpp"
#endif

#define main aai_reference_unused_main
#include REF_SOURCE
#undef main

#include <cstring>
#include <cstdlib>
#include <sstream>

extern "C" {

// mode: 1 = areaAverageInterpolation (Source.cpp:55), 2 = fastAreaAverageInterpolation (Source.cpp:584)
// src: row-major H x W doubles.  On success returns 1, *out is malloc'ed (dH*dW doubles, row-major);
// release with aai_ref_free.  On failure returns 0 and writes the reference's message into err;
// *out is NULL and dW/dH/dIso are left untouched (like the reference leaves dst/dstIsocenter).
int aai_ref_run(int mode, const double *src, int W, int H,
                double srcResX, double srcResY, double dstResX, double dstResY,
                double isoX, double isoY, double angleDeg,
                double **out, int *dW, int *dH, double *dIsoX, double *dIsoY,
                char *err, int errLen)
{
    IMG s, d;
    s.resize((size_t)H);
    for (int y = 0; y < H; ++y) s[y].assign(src + (size_t)y * W, src + (size_t)(y + 1) * W);
    dP dIso = std::make_pair(-12345.0, -54321.0);
    AreaAverageInterpolation aa;

    std::stringstream sink;
    std::streambuf *keep = std::cout.rdbuf(sink.rdbuf());
    std::pair<bool, std::string> r;
    if (mode == 2)
        r = aa.fastAreaAverageInterpolation(s, d, std::make_pair(srcResX, srcResY),
                                            std::make_pair(dstResX, dstResY),
                                            std::make_pair(isoX, isoY), dIso, angleDeg);
    else
        r = aa.areaAverageInterpolation(s, d, std::make_pair(srcResX, srcResY),
                                        std::make_pair(dstResX, dstResY),
                                        std::make_pair(isoX, isoY), dIso, angleDeg);
    std::cout.rdbuf(keep);

    if (out) *out = NULL;
    if (!r.first) {
        if (err && errLen > 0) {
            std::strncpy(err, r.second.c_str(), (size_t)errLen - 1);
            err[errLen - 1] = 0;
        }
        return 0;
    }
    if (err && errLen > 0) err[0] = 0;
    int h = (int)d.size();
    int w = h ? (int)d.front().size() : 0;
    *dW = w; *dH = h; *dIsoX = dIso.first; *dIsoY = dIso.second;
    if (out) {
        double *o = (double *)std::malloc(sizeof(double) * (size_t)(w > 0 ? w : 1) * (size_t)(h > 0 ? h : 1));
        for (int y = 0; y < h; ++y) std::memcpy(o + (size_t)y * w, d[y].data(), sizeof(double) * (size_t)w);
        *out = o;
    }
    return 1;
}

// Empty-image probes for the two "no data" error paths (Source.cpp:123-132): rows==0, or rows>0 with
// an empty first row.
int aai_ref_run_empty(int mode, int rows, char *err, int errLen)
{
    IMG s, d;
    s.resize((size_t)rows);
    dP dIso = std::make_pair(0.0, 0.0);
    AreaAverageInterpolation aa;
    std::stringstream sink;
    std::streambuf *keep = std::cout.rdbuf(sink.rdbuf());
    std::pair<bool, std::string> r = (mode == 2)
        ? aa.fastAreaAverageInterpolation(s, d, std::make_pair(1.0, 1.0), std::make_pair(1.0, 1.0),
                                          std::make_pair(0.0, 0.0), dIso, 0.0)
        : aa.areaAverageInterpolation(s, d, std::make_pair(1.0, 1.0), std::make_pair(1.0, 1.0),
                                      std::make_pair(0.0, 0.0), dIso, 0.0);
    std::cout.rdbuf(keep);
    if (err && errLen > 0) {
        std::strncpy(err, r.second.c_str(), (size_t)errLen - 1);
        err[errLen - 1] = 0;
    }
    return r.first ? 1 : 0;
}

void aai_ref_free(double *p) { std::free(p); }

}  // extern "C"
