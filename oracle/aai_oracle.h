/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the area-average interpolation hot path.
 *
 * Plain-C, double-precision, single-thread restatement of the reference algorithm
 * (/root/reference/Source.cpp:55-1431).  Parity pinned against the unmodified reference
 * (oracle/_ref, built by oracle/Makefile) and the golden vectors in tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * It is never linked into libaai_hip.so and the product path has no fallback onto it.
 */
#ifndef AAI_ORACLE_H
#define AAI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { AAI_ORACLE_MODE_EXACT = 1, AAI_ORACLE_MODE_FAST = 2,        /* Source.cpp:55, :584 */
       AAI_ORACLE_MODE_BILINEAR = 3, AAI_ORACLE_MODE_BICUBIC = 4 }; /* build-defined comparison paths */
enum { AAI_ORACLE_POLICY_REFERENCE = 0, AAI_ORACLE_POLICY_EXACT = 1 };

/* Geometry derived from the arguments (Source.cpp:139-200). */
typedef struct aai_oracle_geom {
    int W, H;                 /* input size */
    unsigned scale;           /* Source.cpp:139 */
    int quadrant;             /* beforehandRotationMode, Source.cpp:140-146 */
    double angle;             /* reduced angle in [0,90) degrees */
    double sn, cs;            /* Source.cpp:147-148 */
    unsigned mW, mH;          /* modSrcSize, Source.cpp:150-156 */
    double isoX, isoY;        /* rescaled isocenter, Source.cpp:173-174 */
    double ratio, side;       /* expansionRatio, dstSideLength, Source.cpp:177-178 */
    unsigned dW, dH;          /* dstSize, Source.cpp:179-180 */
    double dIsoX, dIsoY;      /* integer-valued dstIsocenter, Source.cpp:185-186 */
    double fracX, fracY;      /* dstIsocenterOffset, Source.cpp:183-184 */
    double offX, offY;        /* offset, Source.cpp:187-200 */
    int lt45;                 /* rotationAngle < 45 branch, Source.cpp:230 */
    double tsn, tcs, ttn;     /* tmpSin/tmpCos/tmpTan, Source.cpp:229-240 */
} aai_oracle_geom;

/* Validation + geometry.  Returns 1 on success; on failure returns 0 and copies one of the four
 * reference messages (Source.cpp:115,120,125,130) into err. */
int aai_oracle_geometry(int W, int H, double srcResX, double srcResY, double dstResX, double dstResY,
                        double isoX, double isoY, double angleDeg, aai_oracle_geom *g,
                        char *err, int errLen);

/* Full run.  src: row-major H x W doubles.  *out malloc'ed (dH*dW doubles); free with aai_oracle_free. */
int aai_oracle_run(int mode, int policy, const double *src, int W, int H,
                   double srcResX, double srcResY, double dstResX, double dstResY,
                   double isoX, double isoY, double angleDeg,
                   double **out, int *dW, int *dH, double *dIsoX, double *dIsoY,
                   char *err, int errLen);

/* Rows [row0,row1) of the output only (bounded CPU-baseline samples).  src may be f64 (srcIsF32=0)
 * or f32 (srcIsF32=1, promoted to double per pixel).  out must hold (row1-row0)*dW doubles. */
int aai_oracle_rows(int mode, int policy, const void *src, int srcIsF32, int W, int H,
                    double srcResX, double srcResY, double dstResX, double dstResY,
                    double isoX, double isoY, double angleDeg,
                    int row0, int row1, double *out, char *err, int errLen);

/* A list of n dst pixels (xs[k], ys[k]) -> out[k]: unbiased samples of images too large for a full CPU run.  Each is evaluated
 * right after its predecessor in the reference's loop order, so the loop's persistent state is what the reference would hold. */
int aai_oracle_pixels(int mode, int policy, const void *src, int srcIsF32, int W, int H,
                      double srcResX, double srcResY, double dstResX, double dstResY,
                      double isoX, double isoY, double angleDeg,
                      int n, const int *xs, const int *ys, double *out, char *err, int errLen);

void aai_oracle_free(void *p);

/* Per-pair areas of one dst pixel over the reference's search window (test / debug aid). */
int aai_oracle_pixel_pairs(int policy, int W, int H, double srcResX, double dstResX, double isoX, double isoY,
                           double angleDeg, int dx, int dy, int cap, int *xs, int *ys, double *areas);

/* SURVEY.md Appendix C.1 synthetic image: fp32 uniform [0,1) from a stateless 64-bit hash. */
void aai_oracle_synth_f32(float *dst, int W, int H, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
