/* oracle/orc_graph.c — ORACLE (test infrastructure).  The graph walk of ONNXGraph.swift:95-285 with every
 * converter of Converters.swift that the hot path needs, executing eagerly on the CPU in NCHW fp32.
 *
 * Reference quirks and what this restatement does with each (also tabulated in DESIGN.md):
 *  KEPT   conv out-size ignores dilation (ONNXConvolutionPadding.swift:105-110); Gemm attributes ignored and
 *         weight read as [out,in] (Converters.swift:35-39,228-232); kernel_shape/dilations/strides all
 *         required (:234-237); PRelu slope = first fp32 of raw_data (:372-374); pool uses pads[0],pads[1]
 *         only (:673-674) with the Float-division out-size rule; Upsample scales truncated with Int()
 *         (:499,513-514); N dropped from the input dims, batch supplied at run time (ONNXGraph.swift:204-215);
 *         exactly one graph output (:178-183); Flatten axis != 1 is an error (fatalError mapped to a code);
 *         reference Shape bookkeeping recorded per value (incl. Concat depth*=2, Flatten channels=total);
 *         dilated Conv taps are centred the MPS way: offset = k/2 - pad (ONNXConvolutionPadding.swift:73-79) with tap kh at
 *         centre + (kh - k/2)*d, i.e. the ONNX tap rule with pad' = pad + (k/2)(d-1) on the dilation-ignoring output grid.
 *  SUPERSET (not in the reference registry, ONNXGraph.swift:110-155; named by BASELINE.json north_star): LeakyRelu
 *         (ONNX: alpha attribute, default 0.01) and 2-D MatMul with a constant right operand [K,N].
 *  DROPPED BN epsilon: the attribute IS honoured (default 1e-5) — the reference never reads it so MPS's
 *         default applies, but its own offline flow (ONNX2MPS.py:107 fuse_bn_into_conv) honours it, and the
 *         two flows must agree; fp32 bias is accepted from float_data as well as raw_data (:128-129 reads
 *         raw_data only, yielding an empty bias for typed storage); Concat output C = sum of operand C
 *         (identical to depth*=2 for the only case the reference gets right: two equal operands).
 */
#include "orc.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[256];
const char* orc_last_error(void) { return g_err; }
static int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return code;
}
void orc_free(void* p) { free(p); }

typedef struct {
    orc_bytes name; float* data; int c, h, w; orc_shape ref; int is_input;
} value;

struct orc_graph {
    orc_model* m; orc_config cfg; int mps_flavor;
    value* vals; int n_vals, cap_vals;
    /* name -> tensor (initializers + Constant outputs, ONNXGraph.swift:106-108,259) */
    struct { orc_bytes name; const orc_tensor* t; }* tens; int n_tens, cap_tens;
    int batch;
    char** round_names; int n_round;   /* act_fp16 == 2: round exactly these values */
};

void orc_config_default(orc_config* c) { c->align_corners = 1; c->act_fp16 = 0; for (int i = 0; i < 4; i++) c->dims_override[i] = -1; }

static int beq(orc_bytes a, orc_bytes b) { return a.n == b.n && (a.n == 0 || memcmp(a.p, b.p, a.n) == 0); }
static int bis(orc_bytes a, const char* s) { size_t n = strlen(s); return a.n == n && memcmp(a.p, s, n) == 0; }

static const orc_tensor* find_tensor(const orc_graph* g, orc_bytes name) {
    for (int i = g->n_tens - 1; i >= 0; i--) if (beq(g->tens[i].name, name)) return g->tens[i].t; /* last write wins */
    return NULL;
}
static int add_tensor(orc_graph* g, orc_bytes name, const orc_tensor* t) {
    if (g->n_tens == g->cap_tens) { g->cap_tens = g->cap_tens ? g->cap_tens * 2 : 64;
        g->tens = realloc(g->tens, sizeof(*g->tens) * g->cap_tens); if (!g->tens) return -1; }
    g->tens[g->n_tens].name = name; g->tens[g->n_tens].t = t; g->n_tens++; return 0;
}
static value* find_value(const orc_graph* g, orc_bytes name) {
    for (int i = g->n_vals - 1; i >= 0; i--) if (beq(g->vals[i].name, name)) return &g->vals[i];
    return NULL;
}
static value* put_value(orc_graph* g, orc_bytes name, float* data, int c, int h, int w, orc_shape ref) {
    if (g->n_vals == g->cap_vals) { g->cap_vals = g->cap_vals ? g->cap_vals * 2 : 64;
        g->vals = realloc(g->vals, sizeof(value) * g->cap_vals); if (!g->vals) return NULL; }
    value* v = &g->vals[g->n_vals++];
    v->name = name; v->data = data; v->c = c; v->h = h; v->w = w; v->ref = ref; v->is_input = 0;
    return v;
}
static const orc_attr* find_attr(const orc_node* n, const char* name) {
    for (int i = 0; i < n->n_attr; i++) if (bis(n->attr[i].name, name)) return &n->attr[i];
    return NULL;
}
static float* tensor_floats_alloc(const orc_tensor* t, long* n) {
    long cnt = orc_tensor_floats(t, NULL, 0); if (cnt < 0) { *n = cnt; return NULL; }
    float* f = malloc(sizeof(float) * (cnt ? cnt : 1)); orc_tensor_floats(t, f, cnt); *n = cnt; return f;
}

int orc_graph_create(const void* bytes, size_t n, const orc_config* cfg, orc_graph** out) {
    orc_model* m = NULL;
    int rc = orc_model_decode(bytes, n, &m);
    if (rc) return fail(rc, "protobuf decode failed");
    orc_graph* g = calloc(1, sizeof(*g));
    g->m = m; if (cfg) g->cfg = *cfg; else orc_config_default(&g->cfg);
    g->mps_flavor = bis(m->producer_name, "ONNX2MPS");                        /* ONNXGraph.swift:98-103 */
    for (int i = 0; i < m->n_initializer; i++) add_tensor(g, m->initializer[i].name, &m->initializer[i]); /* :106-108 */
    *out = g; return ORC_OK;
}
static void clear_values(orc_graph* g) {
    for (int i = 0; i < g->n_vals; i++) if (!g->vals[i].is_input) free(g->vals[i].data); else free(g->vals[i].data);
    g->n_vals = 0; g->n_tens = g->m->n_initializer; /* drop Constant-registered tensors */
}
void orc_graph_set_round_list(orc_graph* g, const char* const* names, int n) {
    for (int i = 0; i < g->n_round; i++) free(g->round_names[i]);
    free(g->round_names); g->round_names = NULL; g->n_round = 0;
    if (n <= 0) return;
    g->round_names = malloc(sizeof(char*) * n);
    for (int i = 0; i < n; i++) { g->round_names[i] = malloc(strlen(names[i]) + 1); strcpy(g->round_names[i], names[i]); }
    g->n_round = n; g->cfg.act_fp16 = 2;
}
void orc_graph_destroy(orc_graph* g) {
    if (!g) return; orc_graph_set_round_list(g, NULL, 0); clear_values(g); free(g->vals); free(g->tens); orc_model_free(g->m); free(g);
}
int orc_graph_is_mps_flavor(const orc_graph* g) { return g->mps_flavor; }

int orc_graph_output_shapes(const orc_graph* g, orc_shape* dst, int cap) { /* ONNXGraph.swift:69-91 */
    int k = 0;
    for (int i = 0; i < g->m->n_output; i++) {
        const orc_value_info* o = &g->m->output[i]; orc_shape s;
        if (o->ndims == 3) { s.channels = (int)o->dims[0]; s.width = (int)o->dims[1]; s.height = (int)o->dims[2]; s.depth = 1; }
        else if (o->ndims == 4) { s.channels = (int)o->dims[1]; s.width = (int)o->dims[2]; s.height = (int)o->dims[3]; s.depth = 1; }
        else continue;
        if (k < cap) dst[k] = s; k++;
    }
    return k;
}

static int input_dims(const orc_graph* g, const orc_value_info* vi, int* c, int* h, int* w) { /* :200-215 */
    int64_t d[4];
    if (vi->ndims != 3 && vi->ndims != 4) return fail(ORC_ERR_UNSUPPORTED_INPUT, "input rank %d", vi->ndims);
    for (int i = 0; i < vi->ndims; i++) d[i] = g->cfg.dims_override[i] >= 0 ? g->cfg.dims_override[i] : vi->dims[i];
    if (vi->ndims == 3) { *c = (int)d[0]; *h = (int)d[1]; *w = (int)d[2]; }
    else { *c = (int)d[1]; *h = (int)d[2]; *w = (int)d[3]; }
    return ORC_OK;
}
int orc_graph_input_chw(const orc_graph* g, int* c, int* h, int* w) {
    for (int i = 0; i < g->m->n_input; i++)
        if (!find_tensor(g, g->m->input[i].name)) return input_dims(g, &g->m->input[i], c, h, w);
    return fail(ORC_ERR_UNSUPPORTED_INPUT, "no non-initializer graph input");
}

/* ---- converters ------------------------------------------------------------------------------------ */
#define REF(ch, wd, ht, dp) ((orc_shape){(ch), (wd), (ht), (dp)})

/* fp16 image storage emulation: mode 1 = every node output (MPS stores intermediate images as half),
 * mode 2 = only the values named by orc_graph_set_round_list (the tensors an executor actually materialises) */
static int wants_round(const orc_graph* g, orc_bytes name) {
    if (g->cfg.act_fp16 == 1) return 1;
    if (g->cfg.act_fp16 == 2) { for (int i = 0; i < g->n_round; i++) if (bis(name, g->round_names[i])) return 1; }
    return 0;
}
static int conv_like(orc_graph* g, const orc_node* nd) { /* ConvolutionConverter, Converters.swift:187-338 */
    if (nd->n_input < 1) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Conv: no input");
    value* in = find_value(g, nd->input[0]);
    if (!in) return fail(ORC_ERR_NO_SUCH_OUTPUT, "%.*s: input value missing", (int)nd->op_type.n, nd->op_type.p);
    const orc_tensor* wt = nd->n_input > 1 ? find_tensor(g, nd->input[1]) : NULL;
    if (!wt) return fail(ORC_ERR_INSUFFICIENT_INPUTS, "Conv: weight tensor missing");
    const orc_tensor* bt = nd->n_input > 2 ? find_tensor(g, nd->input[2]) : NULL;
    int is_gemm = bis(nd->op_type, "Gemm"), is_tr = bis(nd->op_type, "ConvTranspose");
    int have_k = 0, have_d = 0, have_s = 0, kh = 0, kw = 0, dh = 1, dw = 1, sh = 1, sw = 1, groups = 1;
    int pt = 0, pl = 0, pb = 0, pr = 0, oph = 0, opw = 0;
    for (int i = 0; i < nd->n_attr; i++) { const orc_attr* a = &nd->attr[i];   /* :208-225 */
        if (bis(a->name, "dilations") && a->n_ints >= 2) { dh = (int)a->ints[0]; dw = (int)a->ints[1]; have_d = 1; }
        else if (bis(a->name, "strides") && a->n_ints >= 2) { sh = (int)a->ints[0]; sw = (int)a->ints[1]; have_s = 1; }
        else if (bis(a->name, "group")) groups = (int)a->i;
        else if (bis(a->name, "pads") && a->n_ints >= 4) { pt = (int)a->ints[0]; pl = (int)a->ints[1]; pb = (int)a->ints[2]; pr = (int)a->ints[3]; }
        else if (bis(a->name, "kernel_shape") && a->n_ints >= 2) { kh = (int)a->ints[0]; kw = (int)a->ints[1]; have_k = 1; }
        else if (bis(a->name, "output_padding") && a->n_ints >= 2) { oph = (int)a->ints[0]; opw = (int)a->ints[1]; }
    }
    if (is_gemm) { kh = kw = dh = dw = sh = sw = 1; have_k = have_d = have_s = 1; }            /* :228-232 */
    if (!have_k || !have_d || !have_s) return fail(ORC_ERR_NOT_ENOUGH_ATTRIBUTES, "Conv: kernel_shape/dilations/strides required"); /* :234-237 */
    if (is_tr && groups != 1) return fail(ORC_ERR_INCONSISTENT_STATE, "ConvTranspose: group=%d is outside the restated subset (group=1)", groups);

    /* ConvDataSource.init :20-138 */
    int OC, IC, wkh, wkw;
    if (is_gemm) { if (wt->ndims < 2) return fail(ORC_ERR_INSUFFICIENT_INPUTS, "Gemm weight rank"); OC = (int)wt->dims[0]; IC = (int)wt->dims[1]; wkh = wkw = 1; }
    else { if (wt->ndims < 4) return fail(ORC_ERR_INSUFFICIENT_INPUTS, "Conv weight rank");
        if (g->mps_flavor) { OC = (int)wt->dims[0]; wkh = (int)wt->dims[1]; wkw = (int)wt->dims[2]; IC = (int)wt->dims[3]; }   /* :40-44 */
        else if (is_tr) { OC = (int)wt->dims[1]; IC = (int)wt->dims[0]; wkh = (int)wt->dims[2]; wkw = (int)wt->dims[3]; }         /* :48-50 [I,O,kH,kW] */
        else { OC = (int)wt->dims[0]; IC = (int)wt->dims[1]; wkh = (int)wt->dims[2]; wkw = (int)wt->dims[3]; } }               /* :46-54 */
    if (wt->data_type != 1 && wt->data_type != 10) return fail(ORC_ERR_UNSUPPORTED_DTYPE, "Conv weight dtype %d", wt->data_type); /* :82-89 .invalid */
    long nw = 0; float* wsrc = tensor_floats_alloc(wt, &nw);
    if (!wsrc) return fail((int)nw, "Conv weight accessor");
    if (nw != (long)OC * IC * wkh * wkw) { free(wsrc); return fail(ORC_ERR_INCONSISTENT_STATE, "Conv weight has %ld elements, dims say %ld", nw, (long)OC * IC * wkh * wkw); }
    float* w = wsrc;
    if (!g->mps_flavor && !is_gemm) { w = malloc(sizeof(float) * nw); orc_reformat_conv_weight(wsrc, w, 4, OC, IC, wkh, wkw, is_tr); free(wsrc); } /* :91-124 */
    float* bias = NULL; long nb = 0;
    if (bt) { bias = tensor_floats_alloc(bt, &nb); if (!bias) { free(w); return fail((int)nb, "Conv bias accessor"); }
        if (nb != OC) { free(w); free(bias); return fail(ORC_ERR_INCONSISTENT_STATE, "Conv bias has %ld elements for %d channels", nb, OC); } }

    int N = g->batch, rc = ORC_OK; float* y; value* ov = NULL; int Ho, Wo;
    if (is_gemm) {
        long in_f = (long)in->c * in->h * in->w;
        if (in_f != IC) { rc = fail(ORC_ERR_INCONSISTENT_STATE, "Gemm: input has %ld features, weight expects %d", in_f, IC); goto done; }
        y = malloc(sizeof(float) * (size_t)N * OC);
        orc_gemm_fc(in->data, w, bias, y, N, IC, OC);
        Ho = Wo = 1;                                                                           /* :323-330 */
    } else {
        int depthwise = (groups != 1) && (groups == OC);                                        /* :57 */
        int cin_total = depthwise ? OC : IC * groups;                                           /* :59-74 */
        if (cin_total != in->c) { rc = fail(ORC_ERR_INCONSISTENT_STATE, "Conv: input C=%d, weight expects %d", in->c, cin_total); goto done; }
        orc_conv_padded_size(in->w, in->h, kh, kw, sh, sw, pt, pl, pb, pr, oph, opw, is_tr, &Wo, &Ho);
        if (Ho <= 0 || Wo <= 0) { rc = fail(ORC_ERR_INCONSISTENT_STATE, "Conv: empty output"); goto done; }
        int ept = pt, epl = pl;                                                                   /* MPS-centred dilated taps (header, KEPT) */
        if (!is_tr) { ept += (kh / 2) * (dh - 1); epl += (kw / 2) * (dw - 1); }
        orc_conv_desc d = {N, in->c, in->h, in->w, OC, kh, kw, sh, sw, dh, dw, ept, epl, pb, pr, groups, Ho, Wo};
        y = malloc(sizeof(float) * (size_t)N * OC * Ho * Wo);
        if (is_tr) orc_conv_transpose2d(&d, in->data, w, bias, y);                              /* :265-287 */
        else orc_conv2d(&d, in->data, w, bias, y);
    }
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, (size_t)N * OC * Ho * Wo);
    for (int i = 0; i < nd->n_output; i++) {  /* addFilter publishes under every output name, ONNXGraph.swift:263-273 */
        float* dat = y; if (i > 0) { dat = malloc(sizeof(float) * (size_t)N * OC * Ho * Wo); memcpy(dat, y, sizeof(float) * (size_t)N * OC * Ho * Wo); }
        ov = put_value(g, nd->output[i], dat, OC, Ho, Wo, REF(1, Wo, Ho, OC));
    }
    (void)ov;
done:
    free(w); free(bias); return rc;
}

static int unary_like(orc_graph* g, const orc_node* nd, int kind) { /* Relu :342-359, PRelu :361-384, BN :797-827 */
    value* in = nd->n_input >= 1 ? find_value(g, nd->input[0]) : NULL;
    if (!in) return fail(ORC_ERR_NO_SUCH_OUTPUT, "%.*s: input value missing", (int)nd->op_type.n, nd->op_type.p);
    size_t cnt = (size_t)g->batch * in->c * in->h * in->w;
    float* y = malloc(sizeof(float) * cnt);
    if (kind == 0) orc_relu(in->data, y, cnt, 0.f);
    else if (kind == 1) {
        const orc_tensor* at = nd->n_input >= 2 ? find_tensor(g, nd->input[1]) : NULL;
        if (!at) { free(y); return fail(ORC_ERR_NO_SUCH_OUTPUT, "PRelu: slope tensor missing"); }
        float a = 0.f; if (at->raw_data.n >= 4) memcpy(&a, at->raw_data.p, 4);                   /* :372-374 */
        orc_relu(in->data, y, cnt, a);
    } else {
        if (nd->n_input < 5) { free(y); return fail(ORC_ERR_INSUFFICIENT_INPUTS, "BatchNormalization: needs 5 inputs"); }
        const orc_tensor* ts[4]; float* p[4] = {0}; long n4[4];
        for (int i = 0; i < 4; i++) { ts[i] = find_tensor(g, nd->input[1 + i]); if (!ts[i]) { free(y); for (int j = 0; j < i; j++) free(p[j]); return fail(ORC_ERR_INSUFFICIENT_INPUTS, "BatchNormalization: tensor %d missing", i + 1); }
            p[i] = tensor_floats_alloc(ts[i], &n4[i]);
            if (!p[i] || n4[i] != in->c) { free(y); for (int j = 0; j <= i; j++) free(p[j]); return fail(ORC_ERR_INCONSISTENT_STATE, "BatchNormalization: parameter %d size", i + 1); } }
        const orc_attr* e = find_attr(nd, "epsilon"); float eps = e ? e->f : 1e-5f;              /* quirk DROPPED, see header */
        orc_batchnorm(in->data, y, g->batch, in->c, in->h * in->w, p[0], p[1], p[2], p[3], eps); /* inputs 1..4 = gamma,beta,mean,var :805-808 */
        for (int i = 0; i < 4; i++) free(p[i]);
    }
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, cnt);
    for (int i = 0; i < nd->n_output; i++) {
        float* dat = y; if (i > 0) { dat = malloc(sizeof(float) * cnt); memcpy(dat, y, sizeof(float) * cnt); }
        put_value(g, nd->output[i], dat, in->c, in->h, in->w, in->ref);
    }
    return ORC_OK;
}

static int add_conv(orc_graph* g, const orc_node* nd) { /* AddConverter :430-451 — both operands must be graph values */
    value* a = nd->n_input >= 2 ? find_value(g, nd->input[0]) : NULL;
    value* b = nd->n_input >= 2 ? find_value(g, nd->input[1]) : NULL;
    if (!a || !b) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Add: operand is not a graph value");
    if (a->c != b->c || a->h != b->h || a->w != b->w) return fail(ORC_ERR_INCONSISTENT_STATE, "Add: shape mismatch (no broadcasting in the reference)");
    size_t cnt = (size_t)g->batch * a->c * a->h * a->w; float* y = malloc(sizeof(float) * cnt);
    orc_add(a->data, b->data, y, cnt);
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, cnt);
    put_value(g, nd->output[0], y, a->c, a->h, a->w, a->ref);
    return ORC_OK;
}

/* Sub :1177, Mul :453, Div :1195 — same guard and shape bookkeeping as Add */
static int binary_conv(orc_graph* g, const orc_node* nd, int op) {
    value* a = nd->n_input >= 2 ? find_value(g, nd->input[0]) : NULL;
    value* b = nd->n_input >= 2 ? find_value(g, nd->input[1]) : NULL;
    if (!a || !b) return fail(ORC_ERR_NO_SUCH_OUTPUT, "%.*s: operand is not a graph value", (int)nd->op_type.n, nd->op_type.p);
    if (a->c != b->c || a->h != b->h || a->w != b->w) return fail(ORC_ERR_INCONSISTENT_STATE, "%.*s: shape mismatch (no broadcasting restated)", (int)nd->op_type.n, nd->op_type.p);
    size_t cnt = (size_t)g->batch * a->c * a->h * a->w; float* y = malloc(sizeof(float) * cnt);
    orc_binary(op, a->data, b->data, y, cnt);
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, cnt);
    put_value(g, nd->output[0], y, a->c, a->h, a->w, a->ref);
    return ORC_OK;
}

/* the MPSCNNNeuron*Node converters with default parameters: Elu :386-409 (alpha attribute required, >= 1 attribute),
 * Exp :411-428, Sigmoid :466-476, Abs :1056, HardSigmoid :1073, Softplus :1090, Softsign :1107, Tanh :1124, Log :1142,
 * Pow :1160 (MPSCNNNeuronPowerNode(source:) = default exponent 1: the identity, kind 0) */
static int neuron_conv(orc_graph* g, const orc_node* nd, int kind) {
    value* in = nd->n_input >= 1 ? find_value(g, nd->input[0]) : NULL;
    if (!in) return fail(ORC_ERR_NO_SUCH_OUTPUT, "%.*s: input value missing", (int)nd->op_type.n, nd->op_type.p);
    float a = 0.f;
    if (kind == 6) { const orc_attr* al = find_attr(nd, "alpha"); if (nd->n_attr < 1 || !al) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Elu: alpha attribute required"); a = al->f; }
    size_t cnt = (size_t)g->batch * in->c * in->h * in->w; float* y = malloc(sizeof(float) * cnt);
    orc_neuron(kind, in->data, y, cnt, a);
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, cnt);
    for (int i = 0; i < nd->n_output; i++) { float* dat = y; if (i > 0) { dat = malloc(sizeof(float) * cnt); memcpy(dat, y, sizeof(float) * cnt); }
        put_value(g, nd->output[i], dat, in->c, in->h, in->w, in->ref); }
    return ORC_OK;
}

/* DropoutConverter :918-939: MPSCNNDropoutNode is the identity at inference; >= 1 attribute and `ratio` required; published
 * under every output name */
static int dropout_conv(orc_graph* g, const orc_node* nd) {
    value* in = nd->n_input >= 1 ? find_value(g, nd->input[0]) : NULL;
    if (!in || nd->n_attr < 1 || !find_attr(nd, "ratio")) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Dropout: input or ratio attribute missing");
    if (nd->n_output < 1) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Dropout: no output");
    size_t cnt = (size_t)g->batch * in->c * in->h * in->w;
    for (int i = 0; i < nd->n_output; i++) { float* dat = malloc(sizeof(float) * cnt); memcpy(dat, in->data, sizeof(float) * cnt);
        put_value(g, nd->output[i], dat, in->c, in->h, in->w, in->ref); }
    return ORC_OK;
}

/* SUPERSET: ONNX LeakyRelu, y = x >= 0 ? x : alpha*x, alpha attribute (default 0.01) */
static int leaky_conv(orc_graph* g, const orc_node* nd) {
    value* in = nd->n_input >= 1 ? find_value(g, nd->input[0]) : NULL;
    if (!in) return fail(ORC_ERR_NO_SUCH_OUTPUT, "LeakyRelu: input value missing");
    const orc_attr* al = find_attr(nd, "alpha"); float a = al ? al->f : 0.01f;
    size_t cnt = (size_t)g->batch * in->c * in->h * in->w; float* y = malloc(sizeof(float) * cnt);
    orc_relu(in->data, y, cnt, a);
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, cnt);
    put_value(g, nd->output[0], y, in->c, in->h, in->w, in->ref);
    return ORC_OK;
}

/* SUPERSET (not in the reference registry, ONNXGraph.swift:110-155): ONNX Clip with lower bound 0 — bounds from the min / max attributes (opset 6) or from
 * constant inputs 1 / 2 (opset 11; an empty name = unbounded); an infinite upper bound is plain ReLU; any other lower bound is refused */
static int clip_conv(orc_graph* g, const orc_node* nd) {
    value* in = nd->n_input >= 1 ? find_value(g, nd->input[0]) : NULL;
    if (!in) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Clip: input value missing");
    float lo = -INFINITY, hi = INFINITY;
    const orc_attr* a;
    if ((a = find_attr(nd, "min"))) lo = a->f;
    if ((a = find_attr(nd, "max"))) hi = a->f;
    for (int k = 1; k <= 2 && k < nd->n_input; k++) {
        if (nd->input[k].n == 0) continue;
        const orc_tensor* t = find_tensor(g, nd->input[k]);
        if (!t) return fail(ORC_ERR_INSUFFICIENT_INPUTS, "Clip: bound must be a constant tensor");
        float v; if (orc_tensor_floats(t, NULL, 0) != 1) return fail(ORC_ERR_INCONSISTENT_STATE, "Clip: bound is not a scalar");
        orc_tensor_floats(t, &v, 1);
        if (k == 1) lo = v; else hi = v;
    }
    if (lo != 0.f) return fail(ORC_ERR_UNKNOWN_NODE_OP_TYPE, "Clip with lower bound %g", (double)lo);
    if (!(hi > 0.f)) return fail(ORC_ERR_INCONSISTENT_STATE, "Clip: upper bound %g", (double)hi);
    size_t cnt = (size_t)g->batch * in->c * in->h * in->w; float* y = malloc(sizeof(float) * cnt);
    if (isinf(hi)) orc_relu(in->data, y, cnt, 0.f); else orc_neuron(12, in->data, y, cnt, hi);
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, cnt);
    put_value(g, nd->output[0], y, in->c, in->h, in->w, in->ref);
    return ORC_OK;
}

/* SUPERSET: ONNX MatMul, 2-D, Y[n][j] = sum_k A[n][k] * B[k][j] with B a constant tensor [K,N] */
static int matmul_conv(orc_graph* g, const orc_node* nd) {
    value* in = nd->n_input >= 2 ? find_value(g, nd->input[0]) : NULL;
    const orc_tensor* bt = nd->n_input >= 2 ? find_tensor(g, nd->input[1]) : NULL;
    if (!in) return fail(ORC_ERR_NO_SUCH_OUTPUT, "MatMul: left operand is not a graph value");
    if (!bt) return fail(ORC_ERR_INSUFFICIENT_INPUTS, "MatMul: right operand must be a constant tensor");
    if (bt->ndims != 2) return fail(ORC_ERR_INCONSISTENT_STATE, "MatMul: rank");
    if (bt->data_type != 1 && bt->data_type != 10) return fail(ORC_ERR_UNSUPPORTED_DTYPE, "MatMul weight dtype %d", bt->data_type);
    long K = (long)bt->dims[0], N = (long)bt->dims[1], nw = 0;
    if ((long)in->c * in->h * in->w != K) return fail(ORC_ERR_INCONSISTENT_STATE, "MatMul: feature count");
    float* b = tensor_floats_alloc(bt, &nw); if (!b) return fail((int)nw, "MatMul weight accessor");
    if (nw != K * N) { free(b); return fail(ORC_ERR_INCONSISTENT_STATE, "MatMul: weight size"); }
    float* y = malloc(sizeof(float) * (size_t)g->batch * N);
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < g->batch; i++)
        for (long j = 0; j < N; j++) { float acc = 0.f; for (long k = 0; k < K; k++) acc += in->data[(size_t)i * K + k] * b[(size_t)k * N + j]; y[(size_t)i * N + j] = acc; }
    free(b);
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, (size_t)g->batch * N);
    put_value(g, nd->output[0], y, (int)N, 1, 1, REF(1, 1, 1, (int)N));
    return ORC_OK;
}

static int softmax_conv(orc_graph* g, const orc_node* nd, int logsm) { /* :697-714, :1213-1232: axis attribute present and == 1 */
    value* in = nd->n_input >= 1 ? find_value(g, nd->input[0]) : NULL;
    const orc_attr* ax = find_attr(nd, "axis");
    if (!in || nd->n_attr < 1 || !ax || ax->i != 1) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Softmax: input missing or axis != 1");
    size_t cnt = (size_t)g->batch * in->c * in->h * in->w; float* y = malloc(sizeof(float) * cnt);
    orc_softmax(in->data, y, g->batch, in->c, in->h * in->w, logsm);
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, cnt);
    put_value(g, nd->output[0], y, in->c, in->h, in->w, in->ref);
    return ORC_OK;
}

/* InstanceNormalization :992-1011 (inputs x, gamma, beta) ; custom_group_norm :1273-1300 (inputs x, groups, gamma, beta).
 * epsilon: the ONNX attribute is honoured (default 1e-5), as for BatchNormalization (quirk dropped, see header). */
static int norm_conv(orc_graph* g, const orc_node* nd, int is_group) {
    const int need = is_group ? 4 : 3;
    value* in = nd->n_input >= need ? find_value(g, nd->input[0]) : NULL;
    const orc_tensor* gt = nd->n_input >= need ? find_tensor(g, nd->input[need - 2]) : NULL;
    const orc_tensor* bt = nd->n_input >= need ? find_tensor(g, nd->input[need - 1]) : NULL;
    const orc_tensor* nt = is_group && nd->n_input >= need ? find_tensor(g, nd->input[1]) : NULL;
    if (!in || !gt || !bt || (is_group && !nt)) return fail(ORC_ERR_NO_SUCH_OUTPUT, "%.*s: input or parameter tensor missing", (int)nd->op_type.n, nd->op_type.p);
    long ng = 0, nb = 0; float* gamma = tensor_floats_alloc(gt, &ng); float* beta = tensor_floats_alloc(bt, &nb);
    int groups = in->c, rc = ORC_OK;
    if (is_group) { int64_t gi = 0; if (orc_tensor_integers(nt, &gi, 1) < 1) { free(gamma); free(beta); return fail(ORC_ERR_NO_SUCH_OUTPUT, "custom_group_norm: groups tensor empty"); } groups = (int)gi; }
    if (!gamma || !beta || ng != in->c || nb != in->c || groups <= 0 || in->c % groups) { free(gamma); free(beta); return fail(ORC_ERR_INCONSISTENT_STATE, "%.*s: parameter size / groups", (int)nd->op_type.n, nd->op_type.p); }
    const orc_attr* e = find_attr(nd, "epsilon"); float eps = e ? e->f : 1e-5f;
    size_t cnt = (size_t)g->batch * in->c * in->h * in->w; float* y = malloc(sizeof(float) * cnt);
    orc_group_norm(in->data, y, g->batch, in->c, in->h * in->w, groups, gamma, beta, eps);
    free(gamma); free(beta);
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, cnt);
    put_value(g, nd->output[0], y, in->c, in->h, in->w, in->ref);
    return rc;
}

static int pad_conv(orc_graph* g, const orc_node* nd) { /* PaddingConverter :944-990: pads = [n,c,h,w begin ; n,c,h,w end] */
    value* in = nd->n_input >= 1 ? find_value(g, nd->input[0]) : NULL;
    const orc_attr* pd = find_attr(nd, "pads");
    if (!in || !pd) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Pad: input or pads attribute missing");
    if (pd->n_ints < 8) return fail(ORC_ERR_INCONSISTENT_STATE, "Pad: 8 pads expected");
    int mode = 0; const orc_attr* ma = find_attr(nd, "mode");
    if (ma && ma->s.n) { if (bis(ma->s, "constant")) mode = 0; else if (bis(ma->s, "reflect")) mode = 1; else if (bis(ma->s, "edge")) mode = 2;
        else return fail(ORC_ERR_INCONSISTENT_STATE, "Pad: mode"); }                                  /* :966-967 */
    if (pd->ints[1] || pd->ints[5] || pd->ints[0] || pd->ints[4]) return fail(ORC_ERR_INCONSISTENT_STATE, "Pad: channel/batch padding is outside the restated subset");
    int pt = (int)pd->ints[2], pl = (int)pd->ints[3], pb = (int)pd->ints[6], pr = (int)pd->ints[7];
    if (pt < 0 || pl < 0 || pb < 0 || pr < 0) return fail(ORC_ERR_INCONSISTENT_STATE, "Pad: negative pads");
    int ho = in->h + pt + pb, wo = in->w + pl + pr;
    size_t cnt = (size_t)g->batch * in->c * ho * wo; float* y = malloc(sizeof(float) * cnt);
    orc_pad2d(in->data, y, g->batch, in->c, in->h, in->w, pt, pl, pb, pr, mode);
    put_value(g, nd->output[0], y, in->c, ho, wo, REF(in->ref.channels, wo, ho, in->ref.depth));   /* :980-985 */
    return ORC_OK;
}

static int pool_conv(orc_graph* g, const orc_node* nd, int is_max) { /* MaxPool :652-695, AveragePool :607-650 */
    value* in = nd->n_input >= 1 ? find_value(g, nd->input[0]) : NULL;
    const orc_attr *ks = find_attr(nd, "kernel_shape"), *pd = find_attr(nd, "pads"), *st = find_attr(nd, "strides");
    if (!in || nd->n_attr < 3 || !ks || !pd || !st || ks->n_ints < 2 || pd->n_ints < 2 || st->n_ints < 2)
        return fail(ORC_ERR_NO_SUCH_OUTPUT, "Pool: kernel_shape, pads and strides are all required");
    int kh = (int)ks->ints[0], kw = (int)ks->ints[1], sh = (int)st->ints[0], sw = (int)st->ints[1];
    int ph = (int)pd->ints[0], pw = (int)pd->ints[1], ho, wo;                                   /* :673-674 */
    orc_pool_padded_size(in->w, in->h, kh, kw, sh, sw, ph, pw, &wo, &ho);
    size_t cnt = (size_t)g->batch * in->c * ho * wo; float* y = malloc(sizeof(float) * cnt);
    if (is_max) orc_maxpool(in->data, y, g->batch, in->c, in->h, in->w, kh, kw, sh, sw, ph, pw, ho, wo);
    else orc_avgpool(in->data, y, g->batch, in->c, in->h, in->w, kh, kw, sh, sw, ph, pw, ho, wo);
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, cnt);
    put_value(g, nd->output[0], y, in->c, ho, wo, REF(in->ref.channels, wo, ho, in->ref.depth));
    return ORC_OK;
}

static int gap_conv(orc_graph* g, const orc_node* nd) { /* GlobalAveragePoolConverter :578-605 */
    value* in = nd->n_input >= 1 ? find_value(g, nd->input[0]) : NULL;
    if (!in) return fail(ORC_ERR_NO_SUCH_OUTPUT, "GlobalAveragePool: input value missing");
    size_t cnt = (size_t)g->batch * in->c; float* y = malloc(sizeof(float) * cnt);
    orc_global_avgpool(in->data, y, g->batch, in->c, in->h * in->w);
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, cnt);
    put_value(g, nd->output[0], y, in->c, 1, 1, REF(in->ref.channels, 1, 1, in->ref.depth));
    return ORC_OK;
}

static int flatten_conv(orc_graph* g, const orc_node* nd) { /* FlattenConverter :879-915 */
    value* in = nd->n_input >= 1 ? find_value(g, nd->input[0]) : NULL;
    const orc_attr* ax = find_attr(nd, "axis");
    if (!in || nd->n_attr < 1 || !ax) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Flatten: input or axis attribute missing");
    if (ax->i != 1) return fail(ORC_ERR_FLATTEN_AXIS, "Flatten: axis %lld (only 1 is supported)", (long long)ax->i); /* :893-896 */
    int total = in->c * in->h * in->w;
    size_t cnt = (size_t)g->batch * total; float* y = malloc(sizeof(float) * cnt);
    memcpy(y, in->data, sizeof(float) * cnt);  /* NCHW order is already the flattened order */
    int ref_total = in->ref.channels * in->ref.width * in->ref.height * in->ref.depth;           /* :901-902 */
    put_value(g, nd->output[0], y, total, 1, 1, REF(ref_total, 1, 1, 1));                        /* :906-911 */
    return ORC_OK;
}

static int reshape_conv(orc_graph* g, const orc_node* nd) { /* ReshapeConverter :830-876 */
    value* in = nd->n_input >= 2 ? find_value(g, nd->input[0]) : NULL;
    const orc_tensor* st = nd->n_input >= 2 ? find_tensor(g, nd->input[1]) : NULL;
    if (!in || !st) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Reshape: input or shape tensor missing");
    int64_t sv[8]; long ns = orc_tensor_integers(st, sv, 8); if (ns < 0) return fail((int)ns, "Reshape: shape dtype");
    int total = in->ref.channels * in->ref.depth * in->ref.height * in->ref.width;               /* :840-843 */
    int src[4] = {in->ref.channels, in->ref.width, in->ref.height, in->ref.depth}, nd3[3];
    for (int i = 0; i < 3; i++) { int d = i < ns ? (int)sv[i] : 1; nd3[i] = d == 0 ? src[i] : d; } /* :844-849 */
    int inf = -1; for (int i = 0; i < 3; i++) if (nd3[i] == -1) { inf = i; break; }
    if (inf >= 0) { int v = total; for (int i = 0; i < 3; i++) if (i != inf) v /= nd3[i]; nd3[inf] = v; } /* :851-857 */
    if ((long)nd3[0] * nd3[1] * nd3[2] != (long)in->c * in->h * in->w) return fail(ORC_ERR_INCONSISTENT_STATE, "Reshape: element count changes");
    size_t cnt = (size_t)g->batch * in->c * in->h * in->w; float* y = malloc(sizeof(float) * cnt);
    memcpy(y, in->data, sizeof(float) * cnt);
    /* (W,H,C) = nd3 (:859-864); data keeps its linear NCHW order */
    put_value(g, nd->output[0], y, nd3[2], nd3[1], nd3[0], REF(in->ref.channels, nd3[0], nd3[1], nd3[2]));
    return ORC_OK;
}

static int upsample_conv(orc_graph* g, const orc_node* nd) { /* UpsampleConverter :478-552 */
    value* in = nd->n_input >= 1 ? find_value(g, nd->input[0]) : NULL;
    if (!in) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Upsample: input value missing");
    const orc_attr *ma = find_attr(nd, "mode"), *sa = find_attr(nd, "scales");
    int have_s = 0, sy = 1, sx = 1;
    if (sa && sa->n_floats >= 4) { sy = (int)sa->floats[2]; sx = (int)sa->floats[3]; have_s = 1; } /* :499 */
    if (!have_s) {
        const orc_tensor* stn = nd->n_input >= 2 ? find_tensor(g, nd->input[1]) : NULL;          /* :505-514 */
        if (!stn || stn->ndims < 1 || stn->dims[0] != 4) return fail(ORC_ERR_NOT_ENOUGH_ATTRIBUTES, "Upsample: scales missing");
        int64_t sv[4]; long ns = orc_tensor_integers(stn, sv, 4);
        if (ns < 0) return fail((int)ns, "Upsample: scales dtype");
        if (ns < 4) return fail(ORC_ERR_NOT_ENOUGH_ATTRIBUTES, "Upsample: scales tensor");
        sy = (int)sv[2]; sx = (int)sv[3];
    }
    if (!ma) return fail(ORC_ERR_NOT_ENOUGH_ATTRIBUTES, "Upsample: mode missing");               /* :517-519 */
    int bil;
    if (bis(ma->s, "nearest")) bil = 0; else if (bis(ma->s, "bilinear") || bis(ma->s, "linear")) bil = 1;
    else return fail(ORC_ERR_UNKNOWN_NODE_OP_TYPE, "Upsample");                                   /* :539-540 */
    int ho = in->h * sy, wo = in->w * sx;
    size_t cnt = (size_t)g->batch * in->c * ho * wo; float* y = malloc(sizeof(float) * cnt);
    orc_upsample(in->data, y, g->batch, in->c, in->h, in->w, sy, sx, bil, g->cfg.align_corners);
    if (wants_round(g, nd->output[0])) orc_round_to_half(y, cnt);
    put_value(g, nd->output[0], y, in->c, ho, wo, REF(in->ref.channels, sx * in->ref.width, sy * in->ref.height, in->ref.depth));
    return ORC_OK;
}

static int concat_conv(orc_graph* g, const orc_node* nd) { /* ConcatConverter :554-574 (axis ignored) */
    if (nd->n_input < 1) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Concat: no inputs");
    const float* xs[16]; int cs[16], ctot = 0;
    if (nd->n_input > 16) return fail(ORC_ERR_INCONSISTENT_STATE, "Concat: >16 operands");
    value* v0 = NULL;
    for (int i = 0; i < nd->n_input; i++) { value* v = find_value(g, nd->input[i]);
        if (!v) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Concat: operand %d is not a graph value", i);
        if (!v0) v0 = v; else if (v->h != v0->h || v->w != v0->w) return fail(ORC_ERR_INCONSISTENT_STATE, "Concat: spatial mismatch");
        xs[i] = v->data; cs[i] = v->c; ctot += v->c; }
    size_t cnt = (size_t)g->batch * ctot * v0->h * v0->w; float* y = malloc(sizeof(float) * cnt);
    orc_concat_c(xs, cs, nd->n_input, y, g->batch, v0->h * v0->w);
    orc_shape r = v0->ref; r.depth *= 2;                                                          /* :566-567 */
    put_value(g, nd->output[0], y, ctot, v0->h, v0->w, r);
    return ORC_OK;
}

static int constant_conv(orc_graph* g, const orc_node* nd) { /* ConstantConverter :716-727 */
    const orc_attr* v = find_attr(nd, "value");
    if (nd->n_output < 1 || !v) return fail(ORC_ERR_NO_SUCH_OUTPUT, "Constant: output or value missing");
    add_tensor(g, nd->output[0], &v->t); return ORC_OK;
}

int orc_graph_run(orc_graph* g, const float* in_nchw, int batch, float** out, int* oc, int* oh, int* ow) {
    clear_values(g); g->batch = batch;
    /* initOutputs, ONNXGraph.swift:197-251 */
    int n_inputs = 0;
    for (int i = 0; i < g->m->n_input; i++) {
        const orc_value_info* vi = &g->m->input[i];
        if (find_tensor(g, vi->name)) continue;                                                   /* :199 */
        int c, h, w, rc = input_dims(g, vi, &c, &h, &w); if (rc) return rc;
        if (n_inputs++) return fail(ORC_ERR_UNSUPPORTED_INPUT, "more than one image input");
        size_t cnt = (size_t)batch * c * h * w; float* x = malloc(sizeof(float) * cnt);
        memcpy(x, in_nchw, sizeof(float) * cnt);
        if (g->cfg.act_fp16) orc_round_to_half(x, cnt);   /* the image itself is fp16 in every storage mode */
        value* v = put_value(g, vi->name, x, c, h, w, REF(1, w, h, c)); v->is_input = 1;          /* :243-248 */
    }
    for (int i = 0; i < g->m->n_node; i++) {                                                      /* :172-176 */
        const orc_node* nd = &g->m->node[i]; orc_bytes op = nd->op_type; int rc;
        if (bis(op, "Conv") || bis(op, "Gemm") || bis(op, "ConvTranspose")) rc = conv_like(g, nd);
        else if (bis(op, "Relu")) rc = unary_like(g, nd, 0);
        else if (bis(op, "PRelu")) rc = unary_like(g, nd, 1);
        else if (bis(op, "BatchNormalization")) rc = unary_like(g, nd, 2);
        else if (bis(op, "Add")) rc = add_conv(g, nd);
        else if (bis(op, "Sub")) rc = binary_conv(g, nd, 1);
        else if (bis(op, "Mul")) rc = binary_conv(g, nd, 2);
        else if (bis(op, "Div")) rc = binary_conv(g, nd, 3);
        else if (bis(op, "Sigmoid")) rc = neuron_conv(g, nd, 3);
        else if (bis(op, "HardSigmoid")) rc = neuron_conv(g, nd, 4);
        else if (bis(op, "Tanh")) rc = neuron_conv(g, nd, 5);
        else if (bis(op, "Elu")) rc = neuron_conv(g, nd, 6);
        else if (bis(op, "Abs")) rc = neuron_conv(g, nd, 7);
        else if (bis(op, "Softplus")) rc = neuron_conv(g, nd, 8);
        else if (bis(op, "Softsign")) rc = neuron_conv(g, nd, 9);
        else if (bis(op, "Exp")) rc = neuron_conv(g, nd, 10);
        else if (bis(op, "Log")) rc = neuron_conv(g, nd, 11);
        else if (bis(op, "Pow")) rc = neuron_conv(g, nd, 0);
        else if (bis(op, "Softmax")) rc = softmax_conv(g, nd, 0);
        else if (bis(op, "LogSoftmax")) rc = softmax_conv(g, nd, 1);
        else if (bis(op, "InstanceNormalization")) rc = norm_conv(g, nd, 0);
        else if (bis(op, "custom_group_norm")) rc = norm_conv(g, nd, 1);
        else if (bis(op, "Pad")) rc = pad_conv(g, nd);
        else if (bis(op, "MaxPool")) rc = pool_conv(g, nd, 1);
        else if (bis(op, "AveragePool")) rc = pool_conv(g, nd, 0);
        else if (bis(op, "GlobalAveragePool")) rc = gap_conv(g, nd);
        else if (bis(op, "Flatten")) rc = flatten_conv(g, nd);
        else if (bis(op, "Reshape")) rc = reshape_conv(g, nd);
        else if (bis(op, "Upsample")) rc = upsample_conv(g, nd);
        else if (bis(op, "Concat")) rc = concat_conv(g, nd);
        else if (bis(op, "Constant")) rc = constant_conv(g, nd);
        else if (bis(op, "Dropout")) rc = dropout_conv(g, nd);
        else if (bis(op, "LeakyRelu")) rc = leaky_conv(g, nd);
        else if (bis(op, "MatMul")) rc = matmul_conv(g, nd);
        else if (bis(op, "Clip")) rc = clip_conv(g, nd);
        else rc = fail(ORC_ERR_UNKNOWN_NODE_OP_TYPE, "%.*s", (int)op.n, op.p);                    /* :173-174 */
        if (rc) return rc;
    }
    if (g->m->n_output != 1) return fail(ORC_ERR_UNSUPPORTED_OUTPUT, "graph has %d outputs", g->m->n_output); /* :178-180 */
    value* o = find_value(g, g->m->output[0].name);
    if (!o) return fail(ORC_ERR_NO_SUCH_OUTPUT, "graph output is not produced by any node");    /* :182-183 */
    size_t cnt = (size_t)batch * o->c * o->h * o->w;
    *out = malloc(sizeof(float) * cnt); memcpy(*out, o->data, sizeof(float) * cnt);
    *oc = o->c; *oh = o->h; *ow = o->w;
    return ORC_OK;
}

const float* orc_graph_value(const orc_graph* g, const char* name, int* c, int* h, int* w) {
    orc_bytes b = {(const uint8_t*)name, strlen(name)}; value* v = find_value(g, b);
    if (!v) return NULL; *c = v->c; *h = v->h; *w = v->w; return v->data;
}
const orc_shape* orc_graph_ref_shape(const orc_graph* g, const char* name) {
    orc_bytes b = {(const uint8_t*)name, strlen(name)}; value* v = find_value(g, b);
    return v ? &v->ref : NULL;
}
