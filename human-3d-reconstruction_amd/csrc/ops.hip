// Plan interpreter + error plumbing of libh3d_hip.so.
#include "common.h"
#include <stdlib.h>

static thread_local char g_err[512] = "";

void h3d_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local char g_kname[160] = "";
static thread_local bool g_dry = false;

bool h3d_note_kernel(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kname, sizeof(g_kname), fmt, ap);
    va_end(ap);
    return g_dry;
}

#ifdef H3D_ABLATE
unsigned long long *h3d_stamp_buffer()
{
    static unsigned long long *buf = [] { void *p = nullptr; (void)hipMalloc(&p, sizeof(unsigned long long) * 65536 * H3D_NSTAMP); return (unsigned long long *)p; }();
    return buf;
}
// profiling builds only (not declared in include/h3d.h): copy the phase stamps of the last launch to the host
extern "C" int h3d_debug_stamps(unsigned long long *dst, int n)
{
    return hipMemcpy(dst, h3d_stamp_buffer(), sizeof(unsigned long long) * n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

int h3d_xcd_mode()
{
    static const int mode = [] { const char *e = getenv("H3D_XCD"); return e ? atoi(e) : 1; }();   // on unless H3D_XCD=0
    return mode;
}

extern "C" const char *h3d_last_error(void) { return g_err; }
extern "C" int h3d_abi_version(void) { return H3D_ABI_VERSION; }
extern "C" int h3d_build_flags(void)
{
    int f = 0;
#ifdef H3D_EXTRA
    f |= H3D_BUILD_EXTRA;
#endif
#ifdef H3D_ABLATE
    f |= H3D_BUILD_ABLATE;
#endif
    return f;
}

int h3d_launch_conv(const h3d_op &op, hipStream_t st);
int h3d_launch_stem(const h3d_op &op, hipStream_t st);
int h3d_launch_elementwise(const h3d_op &op, hipStream_t st);
int h3d_launch_dcn(const h3d_op &op, hipStream_t st);
int h3d_launch_heads(const h3d_op &op, hipStream_t st);
int h3d_launch_dcn2(const h3d_op &op, hipStream_t st);
int h3d_launch_dcn3(const h3d_op &op, hipStream_t st);
int h3d_launch_conv_stream(const h3d_op &op, hipStream_t st);
int h3d_launch_dcn4(const h3d_op &op, hipStream_t st);
int h3d_launch_updcn(const h3d_op &op, hipStream_t st);
int h3d_launch_stem3(const h3d_op &op, hipStream_t st);
int h3d_launch_extra(const h3d_op &op, hipStream_t st);

static int run_one(const h3d_op &op, int i, hipStream_t st);

extern "C" int h3d_op_kernel_name(const h3d_op *op, char *buf, int buflen)
{
    if (!op || !buf || buflen <= 0) H3D_FAIL(H3D_ERR_ARG, "op_kernel_name: null argument");
    g_dry = true;
    g_kname[0] = 0;
    const int rc = run_one(*op, 0, nullptr);
    g_dry = false;
    snprintf(buf, buflen, "%s", g_kname);
    return rc;
}

extern "C" int h3d_run_ops_timed(const h3d_op *ops, int n, void *stream, float *ms)
{
    if (!ops || n <= 0 || !ms) H3D_FAIL(H3D_ERR_ARG, "run_ops_timed: null argument");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t *ev = (hipEvent_t *)malloc(sizeof(hipEvent_t) * (n + 1));
    if (!ev) H3D_FAIL(H3D_ERR_ARG, "run_ops_timed: out of host memory");
    for (int i = 0; i <= n; ++i) (void)hipEventCreate(&ev[i]);
    int rc = H3D_OK;
    (void)hipEventRecord(ev[0], st);
    for (int i = 0; i < n && rc == H3D_OK; ++i) {
        rc = run_one(ops[i], i, st);
        (void)hipEventRecord(ev[i + 1], st);
    }
    if (rc == H3D_OK) {
        (void)hipEventSynchronize(ev[n]);
        for (int i = 0; i < n; ++i) (void)hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
    }
    for (int i = 0; i <= n; ++i) (void)hipEventDestroy(ev[i]);
    free(ev);
    return rc;
}

extern "C" int h3d_run_ops(const h3d_op *ops, int n, void *stream)
{
    if (!ops || n < 0) H3D_FAIL(H3D_ERR_ARG, "run_ops: null plan");
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n; ++i) {
        const int rc = run_one(ops[i], i, st);
        if (rc != H3D_OK) return rc;
    }
    return H3D_OK;
}

static int run_one(const h3d_op &op, int i, hipStream_t st)
{
    int rc;
    if (op.B <= 0 || op.H <= 0 || op.W <= 0 || op.Ho <= 0 || op.Wo <= 0 || op.Cin <= 0 || op.Cout <= 0) {
        h3d_set_error("op %d (kind %d): non-positive dimension", i, op.kind);
        return H3D_ERR_SHAPE;
    }
    switch (op.kind) {
    case H3D_OP_STEM: rc = h3d_launch_stem(op, st); break;
    case H3D_OP_STEM3: rc = h3d_launch_stem3(op, st); break;
    case H3D_OP_CONV: rc = h3d_launch_conv(op, st); break;
    case H3D_OP_CONV_STREAM: rc = h3d_launch_conv_stream(op, st); break;
    case H3D_OP_DCN: rc = h3d_launch_dcn2(op, st); break;
#ifdef H3D_EXTRA
    case H3D_OP_DCN_V1: rc = h3d_launch_dcn(op, st); break;
    case H3D_OP_DCN_FUSED_F16: rc = h3d_launch_dcn4(op, st); break;
    case H3D_OP_UPDCN_F16: rc = h3d_launch_updcn(op, st); break;
#else
    case H3D_OP_DCN_V1:
    case H3D_OP_DCN_FUSED_F16:
    case H3D_OP_UPDCN_F16:
        h3d_set_error("op %d (kind %d): a superseded kernel generation, built only by `make EXTRA=1` (csrc/Makefile)", i, op.kind);
        return H3D_ERR_UNSUPPORTED;
#endif
    case H3D_OP_DCN_FUSED:
    case H3D_OP_DCN_FUSED_STREAM: rc = h3d_launch_dcn3(op, st); break;
    case H3D_OP_HEADS: rc = h3d_launch_heads(op, st); break;
    case H3D_OP_MAXPOOL:
    case H3D_OP_UPADD:
    case H3D_OP_COPY: rc = h3d_launch_elementwise(op, st); break;
    case H3D_OP_IM2COL:
    case H3D_OP_MAXPOOL3:
    case H3D_OP_DEPTH2SPACE: rc = h3d_launch_extra(op, st); break;
    default: h3d_set_error("op %d: unknown kind %d", i, op.kind); return H3D_ERR_ARG;
    }
    if (rc != H3D_OK) {
        char tmp[400];
        snprintf(tmp, sizeof(tmp), "%s", h3d_last_error());
        h3d_set_error("op %d (kind %d): %s", i, op.kind, tmp);
    }
    return rc;
}
