// Plan interpreter + error plumbing of libh3d_hip.so.
#include "common.h"

static thread_local char g_err[512] = "";

void h3d_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *h3d_last_error(void) { return g_err; }
extern "C" int h3d_abi_version(void) { return H3D_ABI_VERSION; }

int h3d_launch_conv(const h3d_op &op, hipStream_t st);
int h3d_launch_stem(const h3d_op &op, hipStream_t st);
int h3d_launch_elementwise(const h3d_op &op, hipStream_t st);
int h3d_launch_dcn(const h3d_op &op, hipStream_t st);

extern "C" int h3d_run_ops(const h3d_op *ops, int n, void *stream)
{
    if (!ops || n < 0) H3D_FAIL(H3D_ERR_ARG, "run_ops: null plan");
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n; ++i) {
        const h3d_op &op = ops[i];
        int rc;
        if (op.B <= 0 || op.H <= 0 || op.W <= 0 || op.Ho <= 0 || op.Wo <= 0 || op.Cin <= 0 || op.Cout <= 0) {
            h3d_set_error("op %d (kind %d): non-positive dimension", i, op.kind);
            return H3D_ERR_SHAPE;
        }
        switch (op.kind) {
        case H3D_OP_STEM: rc = h3d_launch_stem(op, st); break;
        case H3D_OP_CONV: rc = h3d_launch_conv(op, st); break;
        case H3D_OP_DCN: rc = h3d_launch_dcn(op, st); break;
        case H3D_OP_MAXPOOL:
        case H3D_OP_UPADD:
        case H3D_OP_COPY: rc = h3d_launch_elementwise(op, st); break;
        default: h3d_set_error("op %d: unknown kind %d", i, op.kind); return H3D_ERR_ARG;
        }
        if (rc != H3D_OK) {
            char tmp[400];
            snprintf(tmp, sizeof(tmp), "%s", h3d_last_error());
            h3d_set_error("op %d (kind %d): %s", i, op.kind, tmp);
            return rc;
        }
    }
    return H3D_OK;
}
