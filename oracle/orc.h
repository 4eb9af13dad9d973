/* oracle/orc.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of what prisma-ai/Smelter computes on the north-star path
 * (ONNX bytes -> tensor accessors -> weight prep -> node loop -> per-node arithmetic -> readback).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the product
 * (smelter_amd/) never links, imports or executes anything in this directory.
 *
 * PARITY UNPINNED: the reference ships no tests, no golden vectors and no CPU path, and it cannot be
 * compiled or imported here (Swift + Apple MetalPerformanceShaders; ONNX2MPS.py needs the absent `onnx`
 * package) — SURVEY.md §8(c).  What Smelter computes itself (wire decode, typed tensor views, fp16 codec,
 * weight re-layout, size/offset rules, readback packing) follows the cited reference lines and is pinned
 * by hand-derived known-answer tables (tests/test_oracle_kat.py).  What Smelter delegates to Apple MPS
 * (conv/BN/pool/upsample/concat arithmetic) follows the ONNX operator definitions restricted to the
 * attribute subset Smelter reads, cross-checked in the build container against torch.nn.functional on CPU
 * (tools/make_golden.py -> tests/golden/).
 *
 * All activations are NCHW fp32 (deliberately different from the product's NHWC fp16).
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: 1:1 with ONNXGraph.Errors (ONNXGraph.swift:38-47) + decode/abort mappings ---- */
enum {
    ORC_OK = 0,
    ORC_ERR_UNSUPPORTED_INPUT = -1,
    ORC_ERR_UNSUPPORTED_OUTPUT = -2,
    ORC_ERR_UNKNOWN_NODE_OP_TYPE = -3,
    ORC_ERR_NO_SUCH_OUTPUT = -4,
    ORC_ERR_GRAPH_INTERNAL = -5,
    ORC_ERR_INSUFFICIENT_INPUTS = -6,
    ORC_ERR_INCONSISTENT_STATE = -7,
    ORC_ERR_NOT_ENOUGH_ATTRIBUTES = -8,
    ORC_ERR_PROTO_DECODE = -9,
    ORC_ERR_UNSUPPORTED_DTYPE = -10, /* fatalError in Onnx_TensorProto+Extensions.swift:32,60 */
    ORC_ERR_FLATTEN_AXIS = -11       /* fatalError in Converters.swift:895 */
};

/* ---- wire-level structs (onnx.pb.swift field numbers; see orc_proto.c) ---- */
typedef struct { const uint8_t* p; size_t n; } orc_bytes;

typedef struct orc_tensor {
    int64_t* dims; int ndims;
    int32_t data_type;
    orc_bytes name, raw_data;
    float* float_data; size_t n_float;
    int32_t* int32_data; size_t n_int32;
    int64_t* int64_data; size_t n_int64;
    double* double_data; size_t n_double;
    uint64_t* uint64_data; size_t n_uint64;
} orc_tensor;

typedef struct orc_attr {
    orc_bytes name, s;
    float f; int64_t i; int32_t type;
    int has_t; orc_tensor t;
    float* floats; size_t n_floats;
    int64_t* ints; size_t n_ints;
} orc_attr;

typedef struct orc_node {
    orc_bytes* input; int n_input;
    orc_bytes* output; int n_output;
    orc_bytes name, op_type, domain;
    orc_attr* attr; int n_attr;
} orc_node;

typedef struct orc_value_info {
    orc_bytes name; int32_t elem_type;
    int64_t* dims; int ndims;
} orc_value_info;

typedef struct orc_model {
    int64_t ir_version; orc_bytes producer_name, producer_version, graph_name;
    int64_t opset_version;
    orc_node* node; int n_node;
    orc_tensor* initializer; int n_initializer;
    orc_value_info* input; int n_input;
    orc_value_info* output; int n_output;
    uint8_t* owned;   /* private copy of the file bytes all orc_bytes point into */
    void* arena;      /* allocations freed by orc_model_free */
} orc_model;

int  orc_model_decode(const void* bytes, size_t n, orc_model** out);
void orc_model_free(orc_model* m);

/* ---- L2 helpers Smelter computes itself ---- */
/* Float16.swift:17-77 — IEEE binary16 <-> binary32 (widening exact; narrowing RNE, overflow->inf) */
float    orc_half_to_float(uint16_t h);
uint16_t orc_float_to_half(float f);
void     orc_half_to_float_n(const uint16_t* src, float* dst, size_t n);
void     orc_float_to_half_n(const float* src, uint16_t* dst, size_t n);

/* Onnx_TensorProto+Extensions.swift:2-66.  Return element count or a negative ORC_ERR_*; dst may be NULL
 * to query the count. */
long orc_tensor_floats(const orc_tensor* t, float* dst, size_t cap);
long orc_tensor_integers(const orc_tensor* t, int64_t* dst, size_t cap);
long orc_tensor_length(const orc_tensor* t);

/* Array+Extensions.swift:52-93 — elem_size 2 or 4; is_transpose: IOHW -> O,flipH,flipW,I */
void orc_reformat_conv_weight(const void* src, void* dst, int elem_size, int oc, int ic, int kh, int kw,
                              int is_transpose);

/* ONNXConvolutionPadding.swift:91-113 / PyTorchPoolPadding.swift:94-103,56-75 */
void orc_conv_padded_size(int in_w, int in_h, int kh, int kw, int sh, int sw, int pt, int pl, int pb, int pr,
                          int oph, int opw, int is_transpose, int* out_w, int* out_h);
void orc_pool_padded_size(int in_w, int in_h, int kh, int kw, int sh, int sw, int ph, int pw, int* out_w,
                          int* out_h);
void orc_pool_window_start(int in, int out, int k, int s, int* start);

/* MPSImage+Extensions.swift:26-59 — NCHW fp32 -> [image][ceil(C/4) slice][H][W][4 or C(<3)] */
size_t orc_float_array_count(int n, int c, int h, int w);
void   orc_pack_float_array(const float* nchw, int n, int c, int h, int w, float* dst);

/* ---- per-op arithmetic (NCHW fp32; ONNX semantics on the attribute subset Smelter reads) ---- */
typedef struct {
    int N, C, H, W;            /* input */
    int K, kh, kw;             /* K = output channels; weight is OHWI [K][kh][kw][C/groups] */
    int sh, sw, dh, dw;
    int pt, pl, pb, pr;
    int groups;
    int Ho, Wo;
} orc_conv_desc;

void orc_set_acc64(int on); /* conv accumulates in fp64 (spot checks) */
void orc_conv2d(const orc_conv_desc* d, const float* x, const float* w_ohwi, const float* bias, float* y);
/* MPSCNNConvolutionTransposeNode (Converters.swift:265-287) with the data source's flipped OHWI weights
 * (Array+Extensions.swift:71-76): y[n,oc,oy,ox] = b[oc] + sum x[n,ic,iy,ix] * W[ic,oc,kh,kw] over iy*s - pt + kh*d == oy
 * (ONNX ConvTranspose; group 1).  d->Ho/Wo = (in-1)*s - pads + k + output_padding (ONNXConvolutionPadding.swift:97-103). */
void orc_conv_transpose2d(const orc_conv_desc* d, const float* x, const float* w_ohwi_flipped, const float* bias, float* y);
void orc_gemm_fc(const float* x, const float* w_oi, const float* bias, float* y, int n, int in_f, int out_f);
void orc_batchnorm(const float* x, float* y, int n, int c, int hw, const float* gamma, const float* beta,
                   const float* mean, const float* var, float eps);
void orc_relu(const float* x, float* y, size_t n, float slope);
void orc_add(const float* a, const float* b, float* y, size_t n);
/* MPSNNSubtraction/Multiplication/DivisionNode (Converters.swift:1177-1211, :453-464): op 0 add, 1 sub, 2 mul, 3 div */
void orc_binary(int op, const float* a, const float* b, float* y, size_t n);
/* MPSCNNNeuron*Node with default parameters (:386-428, :466-476, :1056-1175): kind 3 sigmoid, 4 hard sigmoid (0.2x+0.5 clamped;
 * attributes not read), 5 tanh, 6 elu(a), 7 abs, 8 softplus, 9 softsign, 10 exp, 11 log — same codes as SM_ACT_* */
void orc_neuron(int kind, const float* x, float* y, size_t n, float a);
/* MPSCNNSoftMaxNode / MPSCNNLogSoftMaxNode (:697-714, :1213-1232): over C for every (n, h, w) */
void orc_softmax(const float* x, float* y, int n, int c, int hw, int log_softmax);
/* MPSCNNInstanceNormalizationNode (:992-1011) / MPSCNNGroupNormalizationNode (:1273-1300): biased variance over HW*(C/groups) */
void orc_group_norm(const float* x, float* y, int n, int c, int hw, int groups, const float* gamma, const float* beta, float eps);
/* MPSNNPadNode (:944-990): mode 0 constant (0), 1 reflect, 2 edge; spatial pads only */
void orc_pad2d(const float* x, float* y, int n, int c, int h, int w, int pt, int pl, int pb, int pr, int mode);
/* Configuration.inputConstraint .forceInputScale (ONNXGraph.swift:219-241): MPSNNLanczosScaleNode (mode 1) / MPSNNBilinearScaleNode
 * (mode 2) in front of the graph.  Apple's scalers are closed source; restated as the textbook resamplers: bilinear with pixel-centre
 * sampling src = (dst + 0.5) in/out - 0.5 clamped to the edge; separable Lanczos-3 widened by max(1, in/out), normalised, edge clamp. */
void orc_resize_nchw(const float* x, float* y, int planes, int h, int w, int ho, int wo, int mode);
void orc_maxpool(const float* x, float* y, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int ph,
                 int pw, int ho, int wo);
void orc_avgpool(const float* x, float* y, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int ph,
                 int pw, int ho, int wo);
void orc_global_avgpool(const float* x, float* y, int n, int c, int hw);
void orc_upsample(const float* x, float* y, int n, int c, int h, int w, int sy, int sx, int bilinear,
                  int align_corners);
void orc_concat_c(const float* const* xs, const int* cs, int k, float* y, int n, int hw);
void orc_round_to_half(float* x, size_t n); /* emulate fp16 image storage (MPS default for intermediates) */

/* ---- graph level (ONNXGraph.swift:95-193 node loop) ---- */
typedef struct {
    int align_corners;         /* Configuration.billinearUpsamplingConfiguration.alignCorners, default 1 */
    int dims_override[4];      /* Configuration.dims: -1 = keep file value */
    int act_fp16;              /* 0: fp32 everywhere; 1: round every node output to fp16 (MPS stores intermediate
                                * images as half); 2: round only the values given to orc_graph_set_round_list */
} orc_config;

typedef struct orc_graph orc_graph;
typedef struct { int channels, width, height, depth; } orc_shape; /* TypeDefinitions.swift:1-33 */

void orc_config_default(orc_config* c);
int  orc_graph_create(const void* bytes, size_t n, const orc_config* cfg, orc_graph** out);
void orc_graph_destroy(orc_graph* g);
/* act_fp16 = 2: emulate fp16 storage for exactly these value names (the tensors an executor materialises) */
void orc_graph_set_round_list(orc_graph* g, const char* const* names, int n);
int  orc_graph_is_mps_flavor(const orc_graph* g);
int  orc_graph_output_shapes(const orc_graph* g, orc_shape* dst, int cap);   /* ONNXGraph.swift:69-91 */
int  orc_graph_input_chw(const orc_graph* g, int* c, int* h, int* w);
/* Runs the node loop on `batch` images; on success *out (malloc'd, caller frees with orc_free) holds the
 * single graph output as NCHW fp32 and oc/oh/ow its per-image dims. */
int  orc_graph_run(orc_graph* g, const float* in_nchw, int batch, float** out, int* oc, int* oh, int* ow);
/* fetch any intermediate value by name after a run (for layer-by-layer debugging); NULL if absent */
const float* orc_graph_value(const orc_graph* g, const char* name, int* c, int* h, int* w);
const orc_shape* orc_graph_ref_shape(const orc_graph* g, const char* name); /* reference bookkeeping */
const char* orc_last_error(void);
void orc_free(void* p);
int  orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
