/* smelter_hip.h — C ABI of libsmelter_hip.so: the MI355X (gfx950) execution layer that replaces what
 * prisma-ai/Smelter delegates to Apple MetalPerformanceShaders.
 *
 * Two levels, both `extern "C"`, plain pointers and sizes only:
 *
 *  (1) KERNEL level  `sm_*`      — one entry point per MPS node class Smelter instantiates
 *      (/root/reference/Sources/Smelter/Converters.swift; the data-source / padding-policy callbacks MPS
 *      makes into Smelter at :140-184, :744-777 and the Padding/ policies become plain descriptor fields).  All
 *      tensors are device pointers; activations are NHWC fp16 with an explicit pixel stride (`ld`, in
 *      elements) so a producer can write straight into a channel slice of a concat buffer; convolution
 *      weights are OHWI fp16 — exactly the layout Smelter's own preprocessing produces
 *      (Extensions/Foundation/Array+Extensions.swift:78-83); bias is fp32 (Converters.swift:126-135).
 *      Every call only ENQUEUES work on `stream` and returns; nothing blocks the host.
 *
 *  (2) GRAPH level   `smelter_*` — the load-and-encode surface of ONNXGraph
 *      (/root/reference/Sources/Smelter/ONNXGraph.swift:95 init(data:configuration:), :169 metalGraph(device:),
 *      README.md:43-44 encode(to:sourceImages:), Extensions/Foundation/MPSImage+Extensions.swift:9 toFloatArray).
 *
 * Return value: 0 on success, a negative SM_ERR_* otherwise; smelter_last_error_string() gives the detail
 * (thread local).  The library never aborts the process: the reference's two fatalError sites
 * (Onnx_TensorProto+Extensions.swift:32,60 and Converters.swift:895) are mapped to error codes.
 */
#ifndef SMELTER_HIP_H
#define SMELTER_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sm_stream_t; /* hipStream_t */

/* ---- error codes: 1:1 with ONNXGraph.Errors (ONNXGraph.swift:38-47) + ours ---- */
enum {
    SM_OK = 0,
    SM_ERR_UNSUPPORTED_INPUT = -1,
    SM_ERR_UNSUPPORTED_OUTPUT = -2,
    SM_ERR_UNKNOWN_NODE_OP_TYPE = -3,
    SM_ERR_NO_SUCH_OUTPUT = -4,
    SM_ERR_GRAPH_INTERNAL = -5,
    SM_ERR_INSUFFICIENT_INPUTS = -6,
    SM_ERR_INCONSISTENT_STATE = -7,
    SM_ERR_NOT_ENOUGH_ATTRIBUTES = -8,
    SM_ERR_PROTO_DECODE = -9,
    SM_ERR_UNSUPPORTED_DTYPE = -10,
    SM_ERR_FLATTEN_AXIS = -11,
    SM_ERR_HIP = -12,
    SM_ERR_INVALID_ARGUMENT = -13
};
const char* smelter_last_error_string(void);
const char* smelter_version(void);

/* ======================================= (1) kernel level ======================================= */

/* Fused / stand-alone neurons.  RELU, PRELU: MPSCNNNeuronReLUNode(a:) Converters.swift:342-384.  The rest are the MPSCNNNeuron*Node
 * classes the reference instantiates with their DEFAULT parameters (Converters.swift:386-428, 466-476, 1056-1175):
 * SIGMOID 1/(1+e^-x); HARD_SIGMOID clamp(0.2x+0.5, 0, 1) (node attributes are not read); TANH; ELU x>=0 ? x : a(e^x-1) with
 * a = `slope`; ABS; SOFTPLUS log(1+e^x); SOFTSIGN x/(1+|x|); EXP e^x; LOG ln x.  (Pow is MPSCNNNeuronPowerNode(source:) with
 * its default exponent 1, i.e. the identity: the compiler aliases it.) */
enum { SM_ACT_NONE = 0, SM_ACT_RELU = 1, SM_ACT_PRELU = 2, SM_ACT_SIGMOID = 3, SM_ACT_HARD_SIGMOID = 4, SM_ACT_TANH = 5, SM_ACT_ELU = 6,
       SM_ACT_ABS = 7, SM_ACT_SOFTPLUS = 8, SM_ACT_SOFTSIGN = 9, SM_ACT_EXP = 10, SM_ACT_LOG = 11,
       SM_ACT_CLIP = 12 /* min(max(x, 0), slope): ONNX Clip with lower bound 0 (ReLU6 = slope 6) — declared superset, not in the reference registry */ };
/* NaN under SM_ACT_RELU is NOT specified: the epilogues that apply ReLU as a packed fp16 max after the rounding (v_pk_max_f16 returns the non-NaN
 * operand: NaN -> 0) and those that apply it as a select on the fp32 sum (NaN stays NaN) coexist, and the autotuner chooses between them per layer.
 * A diverged activation is therefore not guaranteed to surface as NaN in the output; -inf gives 0 everywhere, the sign of a zero result may differ.
 * The reference's MPSCNNNeuronReLUNode (Converters.swift:351) documents neither. */
enum { SM_BIN_ADD = 0, SM_BIN_SUB = 1, SM_BIN_MUL = 2, SM_BIN_DIV = 3 };   /* MPSNNAddition/Subtraction/Multiplication/DivisionNode */
enum { SM_PAD_CONSTANT = 0, SM_PAD_REFLECT = 1, SM_PAD_EDGE = 2 };          /* MPSImageEdgeMode .constant(0) / .mirror / .clamp */

/* Replaces MPSCNNConvolutionDescriptor (Converters.swift:57-79) + ONNX_ConvolutionPadding
 * (Padding/ONNXConvolutionPadding.swift:57-89).  Shapes are resolved when the graph is compiled, not at
 * encode time.  y[n,oy,ox,k] = act(bias[k] + sum x[n, oy*sH-pt+kh*dH, ox*sW-pl+kw*dW, g*Cg+c] * w[k,kh,kw,c] (+ res)) */
typedef struct sm_conv_desc {
    int32_t N, H, W, Cin;         /* input  NHWC */
    int32_t Cout, kH, kW;         /* weight OHWI [Cout][kH][kW][Cin/groups] */
    int32_t sH, sW, dH, dW;
    int32_t pt, pl, pb, pr;       /* pads top,left,bottom,right (pb/pr only enter through Ho/Wo) */
    int32_t groups;               /* 1, or == Cin == Cout (depthwise, Converters.swift:57), or general */
    int32_t Ho, Wo;
    int32_t act; float slope;     /* fused neuron */
    int32_t x_ld, y_ld, res_ld;   /* pixel strides in elements; 0 = dense (Cin / Cout / Cout) */
    /* Output placement (all 0 = dense): output pixel (n, oy, ox) of the Ho x Wo grid is written to pixel
     * (n, oy*y_sH + y_oH, ox*y_sW + y_oW) of an image of y_H x y_W pixels.  This is how a stride-s ConvTranspose runs as
     * s*s ordinary convolutions, one per output phase (smelter_graph_compile); no residual operand in this mode. */
    int32_t y_sH, y_sW, y_oH, y_oW, y_H, y_W;
    /* 1 = `x` is given at HALF resolution ((H/2) x (W/2) pixels, H and W even) and read through a fused nearest-neighbour 2x
     * upsampling: input pixel (iy, ix) of the H x W image is x[n, iy/2, ix/2, :].  MPSCNNUpsamplingNearestNode (Converters.swift:524)
     * in front of a convolution without the upsampled image ever being stored.  3x3 / stride 1 / pad 1, Cin % 64 == 0 only. */
    int32_t x_ups;
} sm_conv_desc;

/* MPSCNNConvolutionNode / MPSCNNFullyConnectedNode (Converters.swift:253, :297).  `residual` (nullable) is an
 * NHWC tensor of the output's shape added before the activation (fuses MPSNNAdditionNode, :440). */
int sm_conv2d_fwd(const sm_conv_desc* d, const void* x, const void* w_ohwi, const float* bias,
                  const void* residual, void* y, sm_stream_t stream);
/* bytes of scratch sm_conv2d_fwd wants for this shape (split-K partials); 0 if none. */
size_t sm_conv2d_workspace_bytes(const sm_conv_desc* d);
int sm_conv2d_fwd_ws(const sm_conv_desc* d, const void* x, const void* w_ohwi, const float* bias,
                     const void* residual, void* y, void* workspace, size_t workspace_bytes, sm_stream_t stream);
/* force a tile configuration (tuning / tests); cfg < 0 restores the heuristic.  Returns the number of configs. */
int sm_conv2d_num_configs(void);
/* label of a tile configuration ("" outside 0 .. num_configs-1): for logs and profile tables only, nothing parses it */
const char* sm_conv2d_config_name(int cfg);
/* > 0 iff `cfg` is a stem-form configuration with loader waves, i.e. one sm_conv2d_stem_f32_fwd_cfg (fp32 NCHW input) takes and
 * sm_conv2d_fwd_cfg (row-pair image input) does not */
int sm_conv2d_config_loader_waves(int cfg);
/* 1 if this shape runs through a tile configuration (so `cfg` of sm_conv2d_fwd_cfg matters and autotuning it makes sense), 0 if a
 * dedicated kernel takes it (depthwise, thin pointwise, direct).  Assumes 16-byte aligned operands. */
int sm_conv2d_uses_tile_config(const sm_conv_desc* d);
int sm_conv2d_fwd_cfg(const sm_conv_desc* d, int cfg, const void* x, const void* w_ohwi, const float* bias,
                      const void* residual, void* y, void* workspace, size_t workspace_bytes, sm_stream_t stream);

/* Two MPSCNNConvolutionNodes joined by an MPSNNAdditionNode (Converters.swift:253, :440) as ONE pointwise convolution over two sources concatenated along K:
 * y = act([W_a | W_b] . [x ; x2] + bias).  `d` = the 1x1 / stride 1 / no padding convolution over x = [N, H, W, K1] with d->Cin = K1 + K2 (the row length of
 * w_ohwi = [Cout][K1 + K2]); x2 = [N, x2_H, x2_W, K2] (pixel stride x2_ld, 0 = K2) is read at pixel (oy * x2_stride, ox * x2_stride) — a stride-2 projection
 * shortcut reads the block input in place.  K1 % 64 == K2 % 64 == 0.  Only the persistent pointwise configurations take a second source. */
int sm_conv2d_dual_fwd_cfg(const sm_conv_desc* d, int cfg, const void* x, const void* x2, int32_t K2, int32_t x2_ld, int32_t x2_H, int32_t x2_W, int32_t x2_stride,
                           const void* w_ohwi, const float* bias, void* y, sm_stream_t stream);

/* MPSCNNConvolutionNode (3x3 / stride 1 / pad 1, Cin, Cout in {32, 64}) + the MPSCNNPoolingMaxNode (2x2 / stride 2, Converters.swift:662)
 * that consumes it, in ONE launch; both the full-resolution tensor `y` (the U-Net's skip connection) and the pooled one are written.
 * W % 32 == 0, H even.  Wider layers (Cin % 64 == 0, any Cout; W % 32 == 0, H % 8 == 0) go to the pooling form of the nine-tap patch kernel.  cfg < 0 picks a configuration. */
int sm_conv2d_pool_supported(const sm_conv_desc* d);
int sm_conv2d_pool_fwd_cfg(const sm_conv_desc* d, int cfg, const void* x, const void* w_ohwi, const float* bias, void* y,
                           void* y_pool, int32_t pool_ld, sm_stream_t stream);

/* ConvolutionConverter (Converters.swift:253) x 2: a 3x3 / stride 1 / pad 1 convolution (Cin, Cout in {32, 64}) whose output feeds a 1x1
 * convolution to n3 <= 32 channels (n3 % 8 == 0) — a segmentation head — in ONE launch: y3 = act3(w3 . act(conv3x3(x)) + b3), the 3x3 output
 * tile being rounded to fp16 exactly as the stored tensor would be.  w3 = [n3][Cout] fp16 (OHWI of a 1x1), b3 fp32 or NULL.  y may be NULL:
 * the 3x3 output is then not stored at all. */
int sm_conv2d_head_supported(const sm_conv_desc* d, int32_t n3);
int sm_conv2d_head_fwd_cfg(const sm_conv_desc* d, int cfg, const void* x, const void* w_ohwi, const float* bias, void* y, const void* w3,
                           const float* b3, int32_t n3, int32_t act3, float slope3, void* y3, int32_t y3_ld, sm_stream_t stream);

/* ConvolutionConverter (Converters.swift:253) x 2 + the MPSNNAdditionNode (:440) and MPSCNNNeuronReLUNode (:351) behind them — the last two
 * layers of a ResNet bottleneck block: 3x3 / stride 1 / pad 1 convolution (Cin = Cout = 64) -> 1x1 convolution to n3 = 32 .. 256 channels
 * (n3 % 32 == 0) + optional residual `res3` ([N,H,W,n3] fp16, row stride res3_ld) + activation, ONE launch:
 * y3 = act3((w3 . act(conv3x3(x)) + b3) + res3), the 3x3 output rounded to fp16 exactly as the stored tensor would be.  y may be NULL. */
int sm_conv2d_xhead_supported(const sm_conv_desc* d, int32_t n3, int32_t has_res);
int sm_conv2d_xhead_fwd_cfg(const sm_conv_desc* d, int cfg, const void* x, const void* w_ohwi, const float* bias, void* y, const void* w3,
                            const float* b3, int32_t n3, int32_t act3, float slope3, const void* res3, int32_t res3_ld, void* y3, int32_t y3_ld,
                            sm_stream_t stream);

/* ... the FIRST block of a stage, whose shortcut is a projection: three ConvolutionConverter nodes (3x3 64 -> 64; 1x1 expand; 1x1 projection of
 * the block input `xs` [N,H,W,64]) + AddConverter + ReluConverter in ONE launch: y3 = act3(w3 . act(conv3x3(x)) + ws . xs + b3) with
 * b3 = the sum of the two 1x1 layers' biases (added by the caller).  The expand layer's own output is never rounded or stored. */
int sm_conv2d_xhead_proj_supported(const sm_conv_desc* d, int32_t n3);
int sm_conv2d_xhead_proj_fwd_cfg(const sm_conv_desc* d, int cfg, const void* x, const void* w_ohwi, const float* bias, const void* w3, const float* b3,
                                 int32_t n3, int32_t act3, float slope3, const void* xs, int32_t xs_ld, const void* ws, void* y3, int32_t y3_ld,
                                 sm_stream_t stream);

/* A whole bottleneck block as ONE tile pass (csrc/kernels/bneck.hip): three ConvolutionConverter nodes (Converters.swift:187-338) — 1x1 reduce C0 -> Cm, 3x3 / stride 1 /
 * pad 1 Cm -> Cm, 1x1 expand Cm -> C3 — each with its ReluConverter (:342-359), and the AddConverter (:430-451) of the identity shortcut:
 *     y = act3(w3 . act2(conv3x3(act1(w1 . x + b1)) + b2) + b3 + res)
 * The two hidden tensors exist in the LDS only, rounded to fp16 there exactly as the layer-by-layer launches round them on their way to memory; x is read once per tile
 * (+ a one-pixel halo) and `res` (NULL: none; normally x itself) once.  Taken: Cm = 64, C0 in {64, 128, 256}, C3 = 32 .. 256 (% 32 == 0), activations of the ReLU family
 * (none / ReLU / leaky with one slope), every tensor below 2 GB.  Weights: w1 [Cm][C0], w2 [Cm][3][3][Cm] (OHWI), w3 [C3][Cm], fp16; biases fp32 or NULL. */
typedef struct sm_bneck_desc {
    int32_t N, H, W;                /* batch, spatial size (the block keeps it) */
    int32_t C0, Cm, C3;             /* channels: block input, hidden (both hidden tensors), block output */
    int32_t x_ld, y_ld, res_ld;     /* pixel strides in elements (res_ld ignored when res == NULL) */
    int32_t act1; float slope1;     /* SM_ACT_* behind the reduce layer */
    int32_t act2; float slope2;     /* ... the 3x3 layer */
    int32_t act3; float slope3;     /* ... the Add */
} sm_bneck_desc;
int sm_bottleneck_supported(const sm_bneck_desc* d);
int sm_bottleneck_fwd(const sm_bneck_desc* d, const void* x, const void* w1, const float* b1, const void* w2_ohwi, const float* b2, const void* w3, const float* b3,
                      const void* res, void* y, sm_stream_t stream);

/* The image stem (first ConvolutionConverter of a network, Cin <= 4) with the texture -> MPSImage conversion INSIDE the launch: loader waves build
 * the stem kernel's row-pair patches from the fp32 NCHW batch (`x_nchw`: [N, C, H_in, W_in]; zero outside the image = the padding (pt, pl)).
 * `d` is the stem-form descriptor sm_conv2d_fwd takes for the same layer (row-pair geometry: Cin = 64, kW = 1, dH = 2, x_ld = 8), `w` the weights
 * packed for it. */
/* ... + the MPSCNNPoolingMaxNode (3x3 / stride 2, Converters.swift:662; window start by PyTorchPoolPadding.swift:60-75 = 0 here) that consumes the stem, in ONE
 * launch (stem_pool.hpp): the stem's own output — the largest tensor of a ResNet — is neither written nor read.  y_pool = [N, pool_H, pool_W, Cout] fp16,
 * pixel stride pool_ld (0 = Cout).  Taken: row-pair stem form with 2 | 4 taps, Cout = 64, and pooled sizes whose last window has its rows / columns 0 and 1 inside the
 * conv image: 2 (pool_H - 1) + 1 < Ho, 2 (pool_W - 1) + 1 < Wo (only row / column 2 of a window may be clamped at the edge). */
int sm_conv2d_stem_f32_pool_supported(const sm_conv_desc* d, int32_t C, int32_t H_in, int32_t W_in, int32_t pool_H, int32_t pool_W);
int sm_conv2d_stem_f32_pool_fwd(const sm_conv_desc* d, const float* x_nchw, int32_t C, int32_t H_in, int32_t W_in, int32_t pt, int32_t pl, const void* w_stem,
                                const float* bias, void* y_pool, int32_t pool_H, int32_t pool_W, int32_t pool_ld, sm_stream_t stream);
int sm_conv2d_stem_f32_supported(const sm_conv_desc* d, int32_t C, int32_t H_in, int32_t W_in);
int sm_conv2d_stem_f32_fwd_cfg(const sm_conv_desc* d, int cfg, const float* x_nchw, int32_t C, int32_t H_in, int32_t W_in, int32_t pt, int32_t pl,
                               const void* w_packed, const float* bias, void* y, sm_stream_t stream);

/* Three MPSCNNConvolutionNodes in a row — [1x1 expand + act1] -> depthwise 3x3 (pad 1, stride 1|2) + act2 -> 1x1 project + act3
 * (+ the block input as residual) — in ONE launch; the two hidden tensors stay in LDS (the "inverted residual" block of
 * MobileNet-v2).  expand = 0: no 1x1 in front (Ch == Cin).  Weights: w1 [Ch][Cin], wd [Ch][3][3] (the depthwise data source's
 * OHWI image), w2 [Cout][Ch], fp16; biases fp32 or NULL.  All channel counts % 8 == 0; activations: the ReLU family (none / ReLU / leaky with slopeN) or SM_ACT_CLIP
 * (min(max(x, 0), slopeN): ReLU6 as a torchvision MobileNet-v2 export holds it).  Four kernels behind it
 * (sm_mbconv_kernel_kind: 1 = all-in-LDS, Cin <= 32, Ch <= 192, Cout <= 64; 2 = chunked; 3 = one wave per tile, Cin <= 46 or weights from
 * global memory up to Cin 110 / Cout 96; 4 = one IMAGE per block for H * W <= 224 at stride 1: Cin % 16 == 0, Cin <= 96 / Cout <= 96 above
 * 64 pixels, Cin <= 160 / Cout <= 384 up to 64 pixels, Ch >= 128, everything within 160 KB of LDS — MobileNet-v2's 14 x 14 and 7 x 7 blocks); sm_mbconv_supported = any of them.
 * Same roundings as the separate launches: the hidden tensors are rounded to fp16 once each. */
typedef struct sm_mbconv_desc {
    int32_t N, H, W, Ho, Wo, stride;
    int32_t Cin, Ch, Cout, expand, residual;
    int32_t act1; float slope1; int32_t act2; float slope2; int32_t act3; float slope3;
    int32_t x_ld, y_ld;           /* pixel strides in elements, 0 = dense */
} sm_mbconv_desc;
int sm_mbconv_supported(const sm_mbconv_desc* d);
int sm_mbconv_kernel_kind(const sm_mbconv_desc* d);   /* 0 = none, else the kernel sm_mbconv_fwd launches for this geometry */
size_t sm_mbconv_lds_bytes(const sm_mbconv_desc* d);
int sm_mbconv_fwd(const sm_mbconv_desc* d, const void* x, const void* w1, const float* b1, const void* wd, const float* bd,
                  const void* w2, const float* b2, void* y, sm_stream_t stream);
/* the same with the kernel named (sm_mbconv_kernel_kind's numbering; 0 = the launcher's choice = sm_mbconv_fwd): a kernel outside whose domain the block lies is not forced — the
 * launcher's choice runs instead.  Tests and tuning; replaces the SMELTER_MBCONV_KERNEL environment switch of rounds 2-4. */
int sm_mbconv_fwd_kind(const sm_mbconv_desc* d, int32_t kind, const void* x, const void* w1, const float* b1, const void* wd, const float* bd,
                       const void* w2, const float* b2, void* y, sm_stream_t stream);

/* The same block directly behind an image stem, the stem included (MobileNet-v2's first three layers): 3x3 convolution of the fp32 NCHW batch
 * (`C` = 3 planes of in_H x in_W, stride st_stride, padding st_pad, activation d->act1) -> depthwise 3x3 -> 1x1 project, ONE launch; neither the
 * input conversion nor the stem's output tensor exists.  d->H x d->W = the stem's output image, d->Ch = its channels (<= 32), d->Cin ignored;
 * w1 = the stem's weights [Ch][3][3][3] (OHWI) fp16.  Same roundings as the separate launches. */
int sm_mbconv_stem_supported(const sm_mbconv_desc* d, int32_t C, int32_t in_H, int32_t in_W, int32_t st_stride, int32_t st_pad);
int sm_mbconv_stem_fwd(const sm_mbconv_desc* d, const float* x_nchw, int32_t C, int32_t in_H, int32_t in_W, int32_t st_stride, int32_t st_pad,
                       const void* w1, const float* b1, const void* wd, const float* bd, const void* w2, const float* b2, void* y,
                       sm_stream_t stream);

/* 3x3 IMAGE STEM: the first MPSCNNConvolutionNode (Converters.swift:253) of a graph whose input is an image with C <= 3 planes, straight from the fp32
 * NCHW batch — no converted copy of the image: y [N, Ho, Wo, Cout] fp16 (pixel stride y_ld halfs, % 8 == 0, 16-byte aligned) =
 * act(conv3x3(x_nchw [N, C, in_H, in_W], w [Cout][3][3][C] fp16 (OHWI), stride 1 | 2, padding `pad` (0 | 1) on all four sides) + bias), Cout <= 64 (% 8 == 0),
 * act in {SM_ACT_NONE, SM_ACT_RELU, leaky / PReLU with one `slope`}.  The image is rounded to fp16 first (as the converted copy was), fp32 accumulation. */
int sm_stem3x3_f32_supported(int32_t N, int32_t C, int32_t in_H, int32_t in_W, int32_t Cout, int32_t stride, int32_t pad);
int sm_stem3x3_f32_fwd(const float* x_nchw, int32_t N, int32_t C, int32_t in_H, int32_t in_W, const void* w_ohwi, const float* bias, int32_t Cout,
                       int32_t stride, int32_t pad, int32_t act, float slope, void* y, int32_t y_ld, sm_stream_t stream);

/* A RUN of MPSCNNConvolutionNodes at one resolution — the bottleneck blocks of a residual stage: 1x1 reduce -> 3x3 -> 1x1 expand (+ MPSNNAdditionNode
 * + MPSCNNNeuronReLUNode; Converters.swift:187-338, :430-451, :342-359) — as ONE persistent launch of 256 blocks (csrc/kernels/stage.hip).  Each phase is the
 * convolution sm_conv2d_fwd (or, with x2 != NULL, sm_conv2d_dual_fwd_cfg) would run on the same operands: 1x1 / stride 1 / no padding or 3x3 / stride 1 /
 * padding 1, groups 1, Cin % 128 == 0, Cout % 8 == 0, images of at most 208 pixels, fp16 NHWC.  An operand of phase i that lies inside the output of an earlier
 * phase of the run is read from it once the producing tiles of the SAME image have published (the batch has no cross-sample dependency: BN uses stored
 * statistics, Converters.swift:760-773); everything else must be complete before the launch.  No phase may overwrite a tensor another phase of the run reads or
 * writes.  The blocks of one XCD form a team that keeps N/8 images for the whole run, so a phase's output is read back from that XCD's L2; when the 256 blocks do not all become
 * resident promptly with 32 per XCD the same launch completes in a slower mode that needs no co-residency (sm_stage_status).
 * `ctl`: sm_stage_ctl_bytes(ph, n) bytes of device memory, zeroed once by the caller; the kernel leaves it zeroed (but for the status words).
 * Same numerics per phase as a phase launched alone through this entry (bit-identical); fp32 accumulation, one rounding to fp16 per phase output. */
#define SM_STAGE_MAX_PHASES 18
typedef struct sm_stage_phase {
    sm_conv_desc d;                 /* d.N = the batch (the same in every phase); d.Cin = K1 + K2 */
    const void* x; const void* w_ohwi; const float* bias; const void* residual; void* y;
    const void* w_packed;           /* optional (NULL: none): the same weights in the run's fragment order (sm_stage_pack_weights), device memory — the kernel then streams them
                                     * as contiguous 1 KB requests instead of 64-byte row pieces (about twice the weight bandwidth per compute unit) */
    const void* x2; int32_t K2, x2_ld, x2_H, x2_W, x2_stride;   /* second source, as in sm_conv2d_dual_fwd_cfg (x2 = NULL: none) */
    int32_t form;                   /* 0 = the planner picks the tile form (fewest bytes through a CU's L1; the 8-wave forms, two waves per SIMD, when every phase of the run has one);
                                       k + 1 = tile form k if it takes the phase (tests, tuning; a run is all 4-wave or all 8-wave forms); < 0 on any phase = the planner keeps to the 4-wave forms */
} sm_stage_phase;
int sm_stage_supported(const sm_stage_phase* ph, int32_t n);
int sm_stage_device_fast(void);   /* 1: the current device has the 256 compute units the static XCD-local mode wants (a smaller / partitioned device still runs a launch correctly, in the slower claimed mode) */
int sm_stage_phase_kind(const sm_stage_phase* ph, int32_t n, int32_t i);   /* tile form phase i runs on (0 .. 8; logs and tests), -1 = run not supported */
size_t sm_stage_ctl_bytes(const sm_stage_phase* ph, int32_t n);
int sm_stage_fwd(const sm_stage_phase* ph, int32_t n, void* ctl, sm_stream_t stream);
/* HOST side, once per model: phase i's OHWI weights (host memory) re-ordered for the tile form the planner gives the phase in THIS run (the phase list and the batch decide it) */
size_t sm_stage_packed_weight_bytes(const sm_stage_phase* ph, int32_t n, int32_t i);
int sm_stage_pack_weights(const sm_stage_phase* ph, int32_t n, int32_t i, const void* w_ohwi_host, void* dst_host);
/* After a run (synchronises `stream`): error_bits 2 = a hand-off wait ran out (results undefined; the graph level turns it into an error of smelter_exec_read_float); mode = the modes the
 * launches since the last call ran in, OR-ed: 1 = static tile assignment, XCD-local hand-off (all 256 blocks arrived promptly, 32 per XCD), 2 = tiles claimed by whichever blocks run, hand-off
 * through agent-scope release / acquire (no co-residency needed: another process on the GPU, a partitioned device).  Clears both words.
 * Test hook: word 12 of `ctl` preset to 2 makes the next launch run in mode 2. */
int sm_stage_status(void* ctl, sm_stream_t stream, int32_t* error_bits, int32_t* mode);

/* MPSCNNConvolutionTransposeNode (Converters.swift:277) + ONNX_ConvolutionPadding(isTranspose) (:66-71, :97-103).
 * `w_ohwi_flipped` is what ConvDataSource hands MPS: out[oc][kH-1-kh][kW-1-kw][ic] = W[ic][oc][kh][kw]
 * (Array+Extensions.swift:71-76).  y[n,oy,ox,oc] = act(bias[oc] + sum x[n,iy,ix,ic] * W[ic,oc,kh,kw]) over
 * iy*sH - pt + kh*dH == oy; d->Ho/Wo = (in-1)*s - pads + k + output_padding; groups must be 1.  Direct gather kernel,
 * any shape; the compiled graph runs stride/dilation-1-decomposable cases as per-phase MFMA convolutions instead. */
int sm_conv_transpose2d_fwd(const sm_conv_desc* d, const void* x, const void* w_ohwi_flipped, const float* bias, void* y,
                            sm_stream_t stream);

/* tuning aid: average duration of `iters` back-to-back launches (HIP events on `stream`) */
int sm_conv2d_time(const sm_conv_desc* d, int cfg, const void* x, const void* w_ohwi, const float* bias, const void* residual,
                   void* y, void* workspace, size_t workspace_bytes, int32_t iters, sm_stream_t stream, float* ms_per_launch);

/* MPSCNNBatchNormalizationNode (Converters.swift:812): y = x*scale[c] + shift[c], scale/shift fp32 folded on
 * the host from gamma,beta,mean,var.  Unfused fallback — the compiler folds BN into the producing conv. */
int sm_batchnorm(const void* x, void* y, int64_t pixels, int32_t C, int32_t x_ld, int32_t y_ld,
                 const float* scale, const float* shift, int32_t act, float slope, sm_stream_t stream);
/* MPSNNAdditionNode (:440) with optional fused neuron; MPSCNNNeuronReLUNode (:351,:376) */
int sm_add(const void* a, const void* b, void* y, int64_t pixels, int32_t C, int32_t a_ld, int32_t b_ld,
           int32_t y_ld, int32_t act, float slope, sm_stream_t stream);
int sm_relu(const void* x, void* y, int64_t pixels, int32_t C, int32_t x_ld, int32_t y_ld, float slope,
            sm_stream_t stream);

/* MPSCNNPoolingMaxNode / MPSCNNPoolingAverageNode + PyTorchPoolPadding (Converters.swift:607-695).
 * y0/x0 = window start of output (0,0) (PyTorchPoolPadding.swift:56-75). */
typedef struct sm_pool_desc {
    int32_t N, H, W, C, kH, kW, sH, sW, y0, x0, Ho, Wo, x_ld, y_ld;
} sm_pool_desc;
int sm_maxpool2d(const sm_pool_desc* d, const void* x, void* y, sm_stream_t stream);
int sm_avgpool2d(const sm_pool_desc* d, const void* x, void* y, sm_stream_t stream);
/* GlobalAveragePoolConverter (:578-605): [N][H*W][C] -> [N][C] */
int sm_global_avgpool(const void* x, void* y, int32_t N, int32_t HW, int32_t C, int32_t x_ld, int32_t y_ld,
                      sm_stream_t stream);

/* MPSCNNUpsamplingNearestNode / MPSCNNUpsamplingBilinearNode(alignCorners) (:522-539), integer scales */
int sm_upsample_nearest(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t sy,
                        int32_t sx, int32_t x_ld, int32_t y_ld, sm_stream_t stream);
int sm_upsample_bilinear(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t sy,
                         int32_t sx, int32_t align_corners, int32_t x_ld, int32_t y_ld, sm_stream_t stream);
/* MPSNNSubtraction/Multiplication/DivisionNode (Converters.swift:1177-1211, :453-464) and Addition: y = act(a op b), same shape */
int sm_binary(int32_t op, const void* a, const void* b, void* y, int64_t pixels, int32_t C, int32_t a_ld, int32_t b_ld,
              int32_t y_ld, int32_t act, float slope, sm_stream_t stream);
/* any SM_ACT_* neuron as a stand-alone pass (the compiler fuses it into the producing conv/add when it can) */
int sm_neuron(const void* x, void* y, int64_t pixels, int32_t C, int32_t x_ld, int32_t y_ld, int32_t act, float slope,
              sm_stream_t stream);
/* MPSCNNSoftMaxNode / MPSCNNLogSoftMaxNode (Converters.swift:697-714, :1213-1232; axis must be 1): over the C channels of a pixel */
int sm_softmax(const void* x, void* y, int64_t pixels, int32_t C, int32_t x_ld, int32_t y_ld, int32_t log_softmax,
               sm_stream_t stream);
/* MPSCNNInstanceNormalizationNode (:992-1011) / MPSCNNGroupNormalizationNode (:1273-1300): per image and group of C/groups
 * channels, y = (x - mean) / sqrt(var + eps) * gamma[c] + beta[c] over H*W*(C/groups) elements (biased variance).
 * groups == C is instance normalisation. */
int sm_group_norm(const void* x, void* y, int32_t N, int32_t HW, int32_t C, int32_t groups, const float* gamma,
                  const float* beta, float eps, int32_t x_ld, int32_t y_ld, sm_stream_t stream);
/* MPSNNPadNode (Converters.swift:944-990): spatial padding, mode SM_PAD_* (constant value 0) */
int sm_pad2d(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad_t, int32_t pad_l,
             int32_t pad_b, int32_t pad_r, int32_t mode, int32_t x_ld, int32_t y_ld, sm_stream_t stream);
/* MPSNNConcatenationNode (:564): strided channel copy; the compiler avoids it by placement when it can */
int sm_copy_channels(const void* x, void* y, int64_t pixels, int32_t C, int32_t x_ld, int32_t y_ld,
                     sm_stream_t stream);

/* MPSNNReshapeNode (Converters.swift:859-864, :898-903) when the reshape is not an alias in NHWC: element order = the linear
 * (c, y, x) order of the source, re-read as [Co][HWo].  Cs*HWs == Co*HWo. */
int sm_reshape_nchw_order(const void* x, void* y, int32_t N, int32_t Cs, int32_t HWs, int32_t x_ld, int32_t Co,
                          int32_t HWo, int32_t y_ld, sm_stream_t stream);

/* Layout/dtype conversion at the boundary (the MPSImage <-> texture step of README.md:33-51).
 * NCHW fp32 -> NHWC fp16 with channels zero-padded to y_ld and a zero spatial halo. */
int sm_nchw_f32_to_nhwc_f16(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t y_ld,
                            int32_t halo_t, int32_t halo_l, int32_t halo_b, int32_t halo_r, sm_stream_t stream);
int sm_nhwc_f16_to_nchw_f32(const void* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t x_ld,
                            sm_stream_t stream);
/* MPSNNBilinearScaleNode / MPSNNLanczosScaleNode (ONNXGraph.swift:222-239) on NCHW fp32 planes.  mode SM_INPUT_FORCE_SCALE_BILINEAR:
 * pixel-centre sampling, src = (dst + 0.5) * in/out - 0.5, clamped to the edge.  SM_INPUT_FORCE_SCALE_LANCZOS: separable
 * Lanczos-3 resampling, support widened by max(1, in/out) when shrinking, weights normalised, edge clamp.  (Apple's scalers are
 * closed source: these are the textbook definitions, pinned in tests to torch bilinear / a numpy Lanczos.) */
int sm_resize_nchw_f32(const float* x, float* y, int32_t planes, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t mode,
                       sm_stream_t stream);
/* First-layer image (C <= 4), NCHW fp32 -> "row-pair image" fp16 [N][Hs][Ws][8]: pixel (y, x) = image pixel
 * (y - pad_t, x - pad_l) in halfs 0..3 and the pixel below it in halfs 4..7, zeros outside the image.  The compiled
 * graph runs the first convolution (ConvolutionConverter, kernel width <= 8) on this layout as a Cin = 64, kW = 1 GEMM. */
int sm_nchw_f32_to_rowpair_f16(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Hs,
                               int32_t Ws, int32_t pad_t, int32_t pad_l, sm_stream_t stream);
/* MPSImage.toFloatArray() element order (MPSImage+Extensions.swift:26-59):
 * [image][ceil(C/4) slice][H][W][4, or C when C < 3] */
int sm_nhwc_f16_to_float_array(const void* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W,
                               int32_t x_ld, sm_stream_t stream);
size_t sm_float_array_count(int32_t N, int32_t C, int32_t H, int32_t W);

/* Host-side weight preparation Smelter does on the CPU at load time */
/* Float16.swift:17-77 (IEEE, RNE) */
void sm_half_to_float(const uint16_t* src, float* dst, size_t n);
void sm_float_to_half(const float* src, uint16_t* dst, size_t n);
/* Array+Extensions.swift:52-93: OIHW -> OHWI, or (is_transpose) IOHW -> O,flipH,flipW,I; elem_size 2|4 */
int sm_reformat_conv_weight(const void* src, void* dst, int32_t elem_size, int32_t oc, int32_t ic, int32_t kh,
                            int32_t kw, int32_t is_transpose);
/* ONNXConvolutionPadding.swift:91-113 / PyTorchPoolPadding.swift:94-103,56-75 */
void sm_conv_padded_size(int32_t in_w, int32_t in_h, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t pt,
                         int32_t pl, int32_t pb, int32_t pr, int32_t oph, int32_t opw, int32_t is_transpose,
                         int32_t* out_w, int32_t* out_h);
void sm_pool_padded_size(int32_t in_w, int32_t in_h, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph,
                         int32_t pw, int32_t* out_w, int32_t* out_h);
int32_t sm_pool_window_start(int32_t in, int32_t out, int32_t k, int32_t s);

/* Device/stream plumbing for callers that do not bring their own HIP runtime binding */
int sm_device_count(int32_t* n);
int sm_set_device(int32_t dev);
int sm_malloc(void** p, size_t bytes);
int sm_free(void* p);
int sm_memcpy_h2d(void* dst, const void* src, size_t bytes, sm_stream_t stream);
int sm_memcpy_d2h(void* dst, const void* src, size_t bytes, sm_stream_t stream);
int sm_memset(void* dst, int value, size_t bytes, sm_stream_t stream);
int sm_stream_create(sm_stream_t* s);
int sm_stream_destroy(sm_stream_t s);
int sm_stream_synchronize(sm_stream_t s);
/* HIP-event timer on `stream` (bench.py measures kernels on the stream they are launched on) */
typedef struct sm_timer sm_timer;
int sm_timer_create(sm_timer** t);
int sm_timer_start(sm_timer* t, sm_stream_t s);
int sm_timer_stop(sm_timer* t, sm_stream_t s);
int sm_timer_elapsed_ms(sm_timer* t, float* ms); /* synchronises on the stop event */
int sm_timer_destroy(sm_timer* t);

/* ======================================== (2) graph level ======================================== */

/* ONNXGraph.Configuration (ONNXGraph.swift:6-36) */
enum { SM_INPUT_NONE = 0, SM_INPUT_FORCE_SCALE_LANCZOS = 1, SM_INPUT_FORCE_SCALE_BILINEAR = 2 };
typedef struct smelter_config {
    int32_t input_constraint;     /* Configuration.InputConstraint: with FORCE_SCALE_* the source image may have any size and is
                                   * resampled to the model's input size first (smelter_exec_encode_sized) */
    int32_t align_corners;        /* billinearUpsamplingConfiguration.alignCorners, default 1 (:20) */
    int32_t dims[4];              /* Configuration.dims: per-axis input-dim override, -1 = keep (:25,200-202) */
    /* --- executor options (no reference counterpart) --- */
    int32_t fuse;                 /* 1 (default): fold BN into conv, fuse Add/Relu epilogues; 0: one kernel per node */
    int32_t use_hip_graph;        /* 1 (default): replay the launch sequence as one hipGraph */
    int32_t autotune;             /* 1: time the conv tile configs per layer at compile; 0 (default): heuristic */
    int32_t keep_intermediates;   /* 1: no activation-buffer reuse, every tensor stays readable (debug / layer parity) */
    int32_t branch_concurrency;   /* 1: a conv only read as a later step's residual (projection shortcut) is hoisted and runs on a
                                   * second stream beside the main branch; 0 (default): strictly one kernel at a time.  Measured on
                                   * ResNet-50 bs=64: the kernels do overlap, but each fork/join pair costs ~16 us of graph-edge latency */
    int32_t fusions;              /* SM_FUSE_* mask of the fusion passes the compiler may apply (with fuse == 1); default SM_FUSE_DEFAULT */
    int32_t autotune_mode;        /* SM_TUNE_IN_SITU (default): a tile configuration is timed behind the layer's two real predecessors;
                                   * SM_TUNE_COLD: behind a 640 MB flush (mis-ranks the streaming kernels) */
    int32_t autotune_max_cfg;     /* 0: every tile configuration; n > 0: only the first n (A/B of library generations on one box) */
    int32_t mbconv_domain[3];     /* {0,0,0}: the default domain of the fused inverted-residual block; else max {Cin, hidden, Cout} it may take */
} smelter_config;
/* Executor behaviour is decided by THIS struct, read once at smelter_graph_create / _compile — no environment variable is consulted after that.
 * smelter_config_default() is where the bench / A/B tools' environment overrides enter (each read there, once): SMELTER_MBCONV=0, SMELTER_XHEAD=0,
 * SMELTER_XHEAD_PROJ=0, SMELTER_KCAT=0|copy, SMELTER_MBCONV_STEM=0, SMELTER_STEM_F32=0, SMELTER_STEM_POOL=0, SMELTER_STEM3=0, SMELTER_STAGE=0|wide, SMELTER_BNECK=0 clear / set the matching SM_FUSE_* bit;
 * SMELTER_AUTOTUNE=cold, SMELTER_AUTOTUNE_MAXCFG=n, SMELTER_MBCONV_DOMAIN="cin,hidden,cout" fill the fields above.  A host that fills the struct itself
 * (the Swift shim of INTEGRATION.md) is not affected by the environment at all.
 * The kernel-level `sm_*` launchers consult NO environment variable (round 5): which kernel runs is a pure function of the descriptor and the explicit arguments
 * (`cfg`, sm_mbconv_fwd_kind's `kind`, sm_stage_phase.form).  The experiment switches of rounds 1-4 (SMELTER_MBCONV_WAVES / _NO_V5, SMELTER_NO_PW_THIN / _PW_THIN_RES /
 * _PW_THIN_K64, SMELTER_DW_STRIP, SM_CONV_ORDER, SM_HX_DEBUG, SM_SP_DEBUG, SM_TIME_COLD, SM_TL_DUMP) exist only in diagnostic builds (`make EXTRA=-DSM_ABLATION`).  Left in the
 * product library, read-only diagnostics that change no kernel choice: SMELTER_ROCTX (trace ranges), SMELTER_DEBUG_SYNC, SMELTER_DEBUG_PLAN (stderr notes). */
enum { SM_FUSE_MBCONV = 1,        /* expand -> depthwise -> project in one launch (mbconv.hip) */
       SM_FUSE_XHEAD = 2,         /* 3x3 -> 1x1 expand (+ residual) in one launch (conv3x3_wsx.hpp) */
       SM_FUSE_XHEAD_PROJ = 4,    /* ... with the projection shortcut as its second GEMM */
       SM_FUSE_KCAT = 8,          /* expand + projection shortcut as ONE pointwise convolution over two sources */
       SM_FUSE_KCAT_COPY = 16,    /* ... through a concatenated copy of the operands instead of reading the second source in place (off by default) */
       SM_FUSE_MBCONV_STEM = 32,  /* image stem inside the first inverted-residual block */
       SM_FUSE_STEM_F32 = 64,     /* input conversion inside the stem launch (loader waves) */
       SM_FUSE_STEM_POOL = 128,   /* stem + max-pool 3x3 / 2 in one launch (stem_pool.hpp) */
       SM_FUSE_STEM3 = 256,       /* a 3x3 stem on <= 3 planes as the staged kernel of stem3.hip (needs SM_FUSE_STEM_F32: it reads the fp32 batch) */
       SM_FUSE_STAGE = 512,       /* runs of consecutive 1x1 / 3x3 stride-1 convolutions at one resolution as ONE persistent launch (sm_stage_fwd, stage.hip) — images of <= 112 pixels */
       SM_FUSE_STAGE_WIDE = 1024, /* ... also at images of 113 .. 208 pixels (the tile forms with 208 accumulator registers: behind the per-layer kernels on ResNet-50's 14 x 14 stage, off by default) */
       SM_FUSE_BNECK = 2048,      /* 1x1 reduce -> 3x3 -> 1x1 expand (+ residual) of a bottleneck block as ONE tile pass, hidden tensors on chip (sm_bottleneck_fwd, bneck.hip; needs SM_FUSE_XHEAD) */
       SM_FUSE_DEFAULT = 1 | 2 | 4 | 8 | 32 | 64 | 128 | 256 | 512 | 2048 };
enum { SM_TUNE_IN_SITU = 0, SM_TUNE_COLD = 1 };
void smelter_config_default(smelter_config* c);

typedef struct smelter_graph smelter_graph; /* ONNXGraph */
typedef struct smelter_exec smelter_exec;   /* what metalGraph(device:) returns (MPSNNGraph) */
typedef struct smelter_shape { int32_t channels, width, height, depth; } smelter_shape; /* TypeDefinitions.swift */

enum { SM_FORMAT_ONNX = 0, SM_FORMAT_MPS_FLAVOR = 1 }; /* ONNXGraph.Format (:49-52) */

/* ONNXGraph(data:configuration:) (:95-156): decodes, detects producer_name=="ONNX2MPS", indexes initializers,
 * registers converters.  The caller may free `onnx` as soon as this returns. */
int smelter_graph_create(const void* onnx, size_t n, const smelter_config* cfg, smelter_graph** out);
void smelter_graph_destroy(smelter_graph* g);
int smelter_graph_format(const smelter_graph* g);                         /* modelFormat (:58) */
int smelter_graph_set_format(smelter_graph* g, int32_t format);           /* it is a public var in the reference */
int smelter_graph_output_shapes(const smelter_graph* g, smelter_shape* dst, int32_t cap); /* outputShapes (:69-91) */
int smelter_graph_num_nodes(const smelter_graph* g);

/* metalGraph(device:) (:169-193): runs the converter loop, folds/casts/re-lays-out weights, plans memory,
 * uploads weights to `hip_device`, captures the launch sequence.  `batch` = MPSImage.numberOfImages the
 * executor is planned for (the reference drops N from the input dims, :204-215).
 * Weight arena: if `weight_arena` is NULL the executor allocates and fills it.  Otherwise the caller owns a
 * device buffer of smelter_exec_weight_bytes() bytes; with `upload_weights`==0 the executor leaves it
 * untouched (batch-shard replicas: rank 0 fills, RCCL broadcasts, the others only bind). */
int smelter_graph_weight_bytes(smelter_graph* g, int32_t batch, size_t* bytes);
/* Host-side image of the weight arena (BN fold + fp16 cast + OHWI + padding, packed); needs no GPU.  Rank 0 of
 * a batch-shard job fills a device buffer with it and RCCL-broadcasts that buffer to the other ranks. */
int smelter_graph_pack_weights(smelter_graph* g, int32_t batch, void* dst, size_t cap);
int smelter_graph_compile(smelter_graph* g, int32_t hip_device, int32_t batch, void* weight_arena,
                          int32_t upload_weights, smelter_exec** out);
/* Execution plan = the tile configuration every step runs with (the autotuner's picks), as int32 words: {'PLAN', tile configurations in the library build,
 * steps, cfg per step}.  A batch-shard job autotunes on rank 0 only and hands the plan to the other ranks with the weight arena (smelter_amd/dist.py), so that all
 * ranks run the IDENTICAL single-GPU executor — per-shard outputs are then bit-identical to the one-GPU run of the same images (the reference has one
 * device, ONNXGraph.swift:169).  get_plan: dst may be NULL to query the size.  default_plan: the plan of a compile WITHOUT autotuning (host only, no GPU).
 * compile_with_plan fails with SM_ERR_INVALID_ARGUMENT if the plan comes from another graph / batch / library build. */
int smelter_exec_get_plan(const smelter_exec* e, int32_t* dst, size_t cap_words, size_t* words);
int smelter_graph_default_plan(smelter_graph* g, int32_t batch, int32_t* dst, size_t cap_words, size_t* words);
int smelter_graph_compile_with_plan(smelter_graph* g, int32_t hip_device, int32_t batch, void* weight_arena, int32_t upload_weights,
                                    const int32_t* plan, size_t plan_words, smelter_exec** out);
void smelter_exec_destroy(smelter_exec* e);

/* encode(to:sourceImages:) (README.md:43-44): enqueue one forward pass on `stream`; asynchronous, no host
 * sync.  in_nchw_f32: device pointer, [batch][C][H][W] fp32.  The result stays in an executor-owned device
 * image (NHWC fp16) until read. */
int smelter_exec_encode(smelter_exec* e, sm_stream_t stream, const float* in_nchw_f32);
/* Configuration.inputConstraint = .forceInputScale (ONNXGraph.swift:219-241: MPSNNLanczosScaleNode / MPSNNBilinearScaleNode in front
 * of the graph): the source images are [batch][C][in_h][in_w] and are resampled to the model's input size on `stream` before the
 * pass.  Fails with SM_ERR_UNSUPPORTED_INPUT when the graph was created with SM_INPUT_NONE and the size differs. */
int smelter_exec_encode_sized(smelter_exec* e, sm_stream_t stream, const float* in_nchw_f32, int32_t in_h, int32_t in_w);
/* toFloatArray (MPSImage+Extensions.swift:9-24): converts the result image to fp32 on `stream`, copies it to
 * host memory and synchronises.  layout 0 = NCHW, 1 = the reference's slice-packed order. */
int smelter_exec_output_dims(const smelter_exec* e, int32_t* n, int32_t* c, int32_t* h, int32_t* w);
int smelter_exec_read_float(smelter_exec* e, sm_stream_t stream, float* dst, size_t cap, int32_t layout);
/* device-side result (NHWC fp16, dense) for callers that keep going on the GPU */
const void* smelter_exec_output_device(const smelter_exec* e);

/* introspection used by tests / bench */
int smelter_exec_num_launches(const smelter_exec* e);       /* kernels per encode after fusion */
int smelter_exec_num_ops(const smelter_exec* e);            /* op records after fusion */
/* Persistent runs of this executor (executor pass 3k; the reference's MPSNNGraph has no counterpart, ONNXGraph.swift:185-190 hands every node to MPS) and what the compile's in-situ
 * A/B measured for each: a host reads here which layers leave in one launch, which form the clock picked and why a run was turned down (bench.py prints it as config.persistent_runs). */
enum { SM_RUN_ON = 0,            /* the run is launched as one kernel */
       SM_RUN_OFF_SLOWER = 1,    /* the A/B measured the run more than 3 % slower than its layers */
       SM_RUN_OFF_TIMEOUT = 2,   /* a hand-off wait ran out during the A/B (results undefined: never kept) */
       SM_RUN_OFF_CLAIMED = 4,   /* no timed launch ran in the static XCD-local mode (the 256 blocks were never co-resident in time) */
       SM_RUN_OFF_DEVICE = 8,    /* the device has fewer than 256 compute units / is partitioned (sm_stage_device_fast) */
       SM_RUN_OFF_PLAN = 16 };   /* an imported plan (smelter_graph_compile_with_plan) carried it turned down */
typedef struct smelter_run_info {
    int32_t first_step, layers;  /* steps [first_step, first_step + layers) of the op list */
    int32_t on;                  /* 1: launched as ONE sm_stage_fwd */
    int32_t timed;               /* 1: the fields below were measured by this compile (autotune on); 0: default / imported decision */
    int32_t samples;             /* timed launches per form (interleaved, medians) */
    int32_t mode_bits;           /* hand-off modes of the timed launches, OR-ed (sm_stage_status: 1 = static XCD-local, 2 = claimed) */
    int32_t err_bits;            /* sm_stage_status error bits seen during the A/B */
    int32_t why_off;             /* SM_RUN_OFF_* bits, 0 when on */
    float t_run_us, t_layers_us; /* median time of the run as one launch / of its layers launched one by one, behind the two real predecessor launches */
} smelter_run_info;
int smelter_exec_num_runs(const smelter_exec* e);
int smelter_exec_run_info(const smelter_exec* e, int32_t idx, smelter_run_info* out);
size_t smelter_exec_weight_bytes(const smelter_exec* e);
void* smelter_exec_weight_arena(const smelter_exec* e);
size_t smelter_exec_activation_bytes(const smelter_exec* e);
double smelter_exec_flops(const smelter_exec* e);           /* 2*MACs of conv+fc per encode */
double smelter_exec_algorithmic_bytes(const smelter_exec* e); /* per-layer compulsory bytes per encode: every layer of the GRAPH charged its own operands */
double smelter_exec_launch_bytes(const smelter_exec* e);      /* post-fusion compulsory bytes per encode: what the launches must read + write (tensors fused away do not count) */
/* per-op description for profiling: returns a static-lifetime string "idx kind name MxNxK cfg" */
const char* smelter_exec_op_desc(const smelter_exec* e, int32_t idx);
/* names of the tensors that exist as fp16 after fusion: the graph input, every kernel output, and the hidden tensors a fused launch rounds to fp16 in
 * registers / LDS without storing them (what a parity oracle must round to emulate the executor).  smelter_exec_value_stored: 1 = the tensor has storage in
 * the activation arena and smelter_exec_read_value can fetch it after an encode; 0 = registers / LDS only (read_value answers SM_ERR_NO_SUCH_OUTPUT). */
int smelter_exec_num_values(const smelter_exec* e);
const char* smelter_exec_value_name(const smelter_exec* e, int32_t idx);
int smelter_exec_value_stored(const smelter_exec* e, int32_t idx);
/* time one op `idx` in isolation (iters launches, HIP events); ms per launch */
int smelter_exec_time_op(smelter_exec* e, sm_stream_t stream, int32_t idx, int32_t iters, float* ms);
/* in-situ per-launch timing: `iters` eager passes over the whole sequence with HIP events on `stream` between
 * consecutive launches.  ms[0] = input conversion, ms[1 + i] = op i; cap >= num_ops + 1. */
int smelter_exec_profile(smelter_exec* e, sm_stream_t stream, const float* in_nchw_f32, int32_t iters, float* ms,
                         int32_t cap);
/* fetch an intermediate value by ONNX name after an encode (debug / layer parity); NCHW fp32 to host */
int smelter_exec_read_value(smelter_exec* e, sm_stream_t stream, const char* name, float* dst, size_t cap,
                            int32_t* c, int32_t* h, int32_t* w);

#ifdef __cplusplus
}
#endif
#endif
