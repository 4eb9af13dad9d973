/* oracle/orc_proto.c — ORACLE (test infrastructure).  proto3 wire decode of the ONNX subset Smelter reads.
 *
 * Follows the generated decoders in /root/reference/Sources/Smelter/onnx.pb.swift:
 *   AttributeProto :1146-1159   NodeProto :1350-1356   ModelProto :1454-1462   GraphProto :1611-1618
 *   TensorProto    :1735-1748   ValueInfoProto :1294-1296   TypeProto/Tensor/Shape/Dim :1870-2014
 *   OperatorSetIdProto :2082-2083
 * Behaviour kept from SwiftProtobuf (the un-vendored dependency, >=1.18.0 <2.0.0, Package.swift:18-22):
 * repeated scalars are accepted packed or unpacked; unknown fields are skipped by wire type; a truncated
 * varint / length is a decode error.
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

typedef struct blk { struct blk* next; } blk;
typedef struct { const uint8_t* p; const uint8_t* end; int err; } rd;

static void* aalloc(orc_model* m, size_t n) {
    blk* b = (blk*)calloc(1, sizeof(blk) + (n ? n : 1));
    if (!b) return NULL;
    b->next = (blk*)m->arena; m->arena = b;
    return (void*)(b + 1);
}

static uint64_t rvarint(rd* r) {
    uint64_t v = 0; int shift = 0;
    while (r->p < r->end && shift < 64) {
        uint8_t b = *r->p++;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return v;
        shift += 7;
    }
    r->err = 1; return 0;
}
static rd rsub(rd* r) { /* length-delimited payload */
    rd s = {r->p, r->p, 0};
    uint64_t n = rvarint(r);
    if (r->err || n > (uint64_t)(r->end - r->p)) { r->err = 1; s.err = 1; return s; }
    s.p = r->p; s.end = r->p + n; r->p += n;
    return s;
}
static orc_bytes rbytes(rd* r) { rd s = rsub(r); orc_bytes b = {s.p, (size_t)(s.end - s.p)}; if (s.err) { b.p = NULL; b.n = 0; } return b; }
static uint32_t rfix32(rd* r) { uint32_t v = 0; if (r->end - r->p < 4) { r->err = 1; return 0; } memcpy(&v, r->p, 4); r->p += 4; return v; }
static uint64_t rfix64(rd* r) { uint64_t v = 0; if (r->end - r->p < 8) { r->err = 1; return 0; } memcpy(&v, r->p, 8); r->p += 8; return v; }
static void rskip(rd* r, int wt) {
    switch (wt) {
    case 0: (void)rvarint(r); break;
    case 1: (void)rfix64(r); break;
    case 2: (void)rsub(r); break;
    case 5: (void)rfix32(r); break;
    default: r->err = 1; /* groups are not used by ONNX */
    }
}

/* growable typed vectors living in the model arena ------------------------------------------- */
#define VEC(T) struct { T* v; size_t n, cap; }
#define VPUSH(m, vec, T, val) do { \
    if ((vec).n == (vec).cap) { size_t nc = (vec).cap ? (vec).cap * 2 : 8; \
        T* nv = (T*)aalloc((m), nc * sizeof(T)); if (!nv) { return -1; } \
        if ((vec).n) memcpy(nv, (vec).v, (vec).n * sizeof(T)); (vec).v = nv; (vec).cap = nc; } \
    (vec).v[(vec).n++] = (val); } while (0)

/* repeated scalar of wire type varint (packed or not) */
static int rep_varint64(orc_model* m, rd* r, int wt, int64_t** v, size_t* n, size_t* cap) {
    VEC(int64_t) vec = {*v, *n, *cap};
    if (wt == 2) { rd s = rsub(r); if (s.err) return -1; while (s.p < s.end && !s.err) VPUSH(m, vec, int64_t, (int64_t)rvarint(&s)); if (s.err) r->err = 1; }
    else if (wt == 0) VPUSH(m, vec, int64_t, (int64_t)rvarint(r));
    else r->err = 1;
    *v = vec.v; *n = vec.n; *cap = vec.cap; return 0;
}
static int rep_fixed(orc_model* m, rd* r, int wt, int width, void** v, size_t* n) {
    /* repeated float (wt 5) / double (wt 1), packed (wt 2) or not; re-allocates exactly (rare unpacked) */
    size_t add; const uint8_t* src; uint8_t tmp[8];
    if (wt == 2) { rd s = rsub(r); if (s.err) return -1; add = (size_t)(s.end - s.p) / width; src = s.p; }
    else if ((wt == 5 && width == 4) || (wt == 1 && width == 8)) {
        if (r->end - r->p < width) { r->err = 1; return -1; }
        memcpy(tmp, r->p, width); r->p += width; add = 1; src = tmp; }
    else { r->err = 1; return -1; }
    uint8_t* nv = (uint8_t*)aalloc(m, (*n + add) * width);
    if (!nv) return -1;
    if (*n) memcpy(nv, *v, *n * width);
    memcpy(nv + *n * width, src, add * width);
    *v = nv; *n += add; return 0;
}

/* TensorProto: onnx.pb.swift:1735-1748 ---------------------------------------------------------- */
static int dec_tensor(orc_model* m, rd s, orc_tensor* t) {
    size_t dcap = 0, dn = 0, c32 = 0, c64 = 0, cu64 = 0;
    int64_t* i32tmp = NULL; size_t n32 = 0;
    int64_t* u64tmp = NULL; size_t nu64 = 0;
    memset(t, 0, sizeof(*t));
    while (s.p < s.end && !s.err) {
        uint64_t key = rvarint(&s); int f = (int)(key >> 3), wt = (int)(key & 7);
        switch (f) {
        case 1: if (rep_varint64(m, &s, wt, &t->dims, &dn, &dcap)) return -1; break;
        case 2: t->data_type = (int32_t)rvarint(&s); break;
        case 4: if (rep_fixed(m, &s, wt, 4, (void**)&t->float_data, &t->n_float)) return -1; break;
        case 5: if (rep_varint64(m, &s, wt, &i32tmp, &n32, &c32)) return -1; break;
        case 7: if (rep_varint64(m, &s, wt, &t->int64_data, &t->n_int64, &c64)) return -1; break;
        case 8: t->name = rbytes(&s); break;
        case 9: t->raw_data = rbytes(&s); break;
        case 10: if (rep_fixed(m, &s, wt, 8, (void**)&t->double_data, &t->n_double)) return -1; break;
        case 11: if (rep_varint64(m, &s, wt, &u64tmp, &nu64, &cu64)) return -1; break;
        default: rskip(&s, wt);
        }
    }
    if (s.err) return -1;
    t->ndims = (int)dn;
    if (n32) { t->int32_data = (int32_t*)aalloc(m, n32 * 4); if (!t->int32_data) return -1;
        for (size_t i = 0; i < n32; i++) t->int32_data[i] = (int32_t)i32tmp[i]; t->n_int32 = n32; }
    if (nu64) { t->uint64_data = (uint64_t*)u64tmp; t->n_uint64 = nu64; }
    return 0;
}

/* AttributeProto: onnx.pb.swift:1146-1159 */
static int dec_attr(orc_model* m, rd s, orc_attr* a) {
    size_t icap = 0;
    memset(a, 0, sizeof(*a));
    while (s.p < s.end && !s.err) {
        uint64_t key = rvarint(&s); int f = (int)(key >> 3), wt = (int)(key & 7);
        switch (f) {
        case 1: a->name = rbytes(&s); break;
        case 2: { uint32_t u = rfix32(&s); memcpy(&a->f, &u, 4); } break;
        case 3: a->i = (int64_t)rvarint(&s); break;
        case 4: a->s = rbytes(&s); break;
        case 5: { rd ts = rsub(&s); if (ts.err || dec_tensor(m, ts, &a->t)) return -1; a->has_t = 1; } break;
        case 7: if (rep_fixed(m, &s, wt, 4, (void**)&a->floats, &a->n_floats)) return -1; break;
        case 8: if (rep_varint64(m, &s, wt, &a->ints, &a->n_ints, &icap)) return -1; break;
        case 20: a->type = (int32_t)rvarint(&s); break;
        default: rskip(&s, wt);
        }
    }
    return s.err ? -1 : 0;
}

/* NodeProto: onnx.pb.swift:1350-1356 */
static int dec_node(orc_model* m, rd s, orc_node* nd) {
    VEC(orc_bytes) in = {0}, out = {0}; VEC(orc_attr) at = {0};
    memset(nd, 0, sizeof(*nd));
    while (s.p < s.end && !s.err) {
        uint64_t key = rvarint(&s); int f = (int)(key >> 3), wt = (int)(key & 7);
        switch (f) {
        case 1: { orc_bytes b = rbytes(&s); VPUSH(m, in, orc_bytes, b); } break;
        case 2: { orc_bytes b = rbytes(&s); VPUSH(m, out, orc_bytes, b); } break;
        case 3: nd->name = rbytes(&s); break;
        case 4: nd->op_type = rbytes(&s); break;
        case 5: { rd as = rsub(&s); orc_attr a; if (as.err || dec_attr(m, as, &a)) return -1; VPUSH(m, at, orc_attr, a); } break;
        case 7: nd->domain = rbytes(&s); break;
        default: rskip(&s, wt);
        }
    }
    nd->input = in.v; nd->n_input = (int)in.n; nd->output = out.v; nd->n_output = (int)out.n;
    nd->attr = at.v; nd->n_attr = (int)at.n;
    return s.err ? -1 : 0;
}

/* ValueInfoProto{1 name, 2 type{1 tensor_type{1 elem_type, 2 shape{1 dim{1 dim_value,2 dim_param}}}}} */
static int dec_value_info(orc_model* m, rd s, orc_value_info* vi) {
    VEC(int64_t) dims = {0};
    memset(vi, 0, sizeof(*vi));
    while (s.p < s.end && !s.err) {
        uint64_t key = rvarint(&s); int f = (int)(key >> 3), wt = (int)(key & 7);
        if (f == 1) vi->name = rbytes(&s);
        else if (f == 2) {
            rd ty = rsub(&s);
            while (ty.p < ty.end && !ty.err) {
                uint64_t k2 = rvarint(&ty); int f2 = (int)(k2 >> 3), w2 = (int)(k2 & 7);
                if (f2 == 1) { /* tensor_type */
                    rd tt = rsub(&ty);
                    while (tt.p < tt.end && !tt.err) {
                        uint64_t k3 = rvarint(&tt); int f3 = (int)(k3 >> 3), w3 = (int)(k3 & 7);
                        if (f3 == 1) vi->elem_type = (int32_t)rvarint(&tt);
                        else if (f3 == 2) { /* shape */
                            rd sh = rsub(&tt);
                            while (sh.p < sh.end && !sh.err) {
                                uint64_t k4 = rvarint(&sh); int f4 = (int)(k4 >> 3), w4 = (int)(k4 & 7);
                                if (f4 == 1) { /* dim */
                                    rd dm = rsub(&sh); int64_t dv = 0; /* dim_param (symbolic) reads as 0 */
                                    while (dm.p < dm.end && !dm.err) {
                                        uint64_t k5 = rvarint(&dm); int f5 = (int)(k5 >> 3), w5 = (int)(k5 & 7);
                                        if (f5 == 1) dv = (int64_t)rvarint(&dm); else rskip(&dm, w5);
                                    }
                                    if (dm.err) sh.err = 1;
                                    VPUSH(m, dims, int64_t, dv);
                                } else rskip(&sh, w4);
                            }
                            if (sh.err) tt.err = 1;
                        } else rskip(&tt, w3);
                    }
                    if (tt.err) ty.err = 1;
                } else rskip(&ty, w2);
            }
            if (ty.err) s.err = 1;
        } else rskip(&s, wt);
    }
    vi->dims = dims.v; vi->ndims = (int)dims.n;
    return s.err ? -1 : 0;
}

/* GraphProto: onnx.pb.swift:1611-1618 */
static int dec_graph(orc_model* m, rd s) {
    VEC(orc_node) nodes = {0}; VEC(orc_tensor) inits = {0}; VEC(orc_value_info) ins = {0}, outs = {0};
    while (s.p < s.end && !s.err) {
        uint64_t key = rvarint(&s); int f = (int)(key >> 3), wt = (int)(key & 7);
        switch (f) {
        case 1: { rd ns = rsub(&s); orc_node nd; if (ns.err || dec_node(m, ns, &nd)) return -1; VPUSH(m, nodes, orc_node, nd); } break;
        case 2: m->graph_name = rbytes(&s); break;
        case 5: { rd ts = rsub(&s); orc_tensor t; if (ts.err || dec_tensor(m, ts, &t)) return -1; VPUSH(m, inits, orc_tensor, t); } break;
        case 11: { rd vs = rsub(&s); orc_value_info v; if (vs.err || dec_value_info(m, vs, &v)) return -1; VPUSH(m, ins, orc_value_info, v); } break;
        case 12: { rd vs = rsub(&s); orc_value_info v; if (vs.err || dec_value_info(m, vs, &v)) return -1; VPUSH(m, outs, orc_value_info, v); } break;
        default: rskip(&s, wt);
        }
    }
    m->node = nodes.v; m->n_node = (int)nodes.n; m->initializer = inits.v; m->n_initializer = (int)inits.n;
    m->input = ins.v; m->n_input = (int)ins.n; m->output = outs.v; m->n_output = (int)outs.n;
    return s.err ? -1 : 0;
}

/* ModelProto: onnx.pb.swift:1454-1462 */
int orc_model_decode(const void* bytes, size_t n, orc_model** out) {
    orc_model* m = (orc_model*)calloc(1, sizeof(orc_model));
    if (!m) return ORC_ERR_PROTO_DECODE;
    m->owned = (uint8_t*)malloc(n ? n : 1);
    if (!m->owned) { free(m); return ORC_ERR_PROTO_DECODE; }
    memcpy(m->owned, bytes, n);
    rd r = {m->owned, m->owned + n, 0};
    int bad = 0;
    while (r.p < r.end && !r.err && !bad) {
        uint64_t key = rvarint(&r); int f = (int)(key >> 3), wt = (int)(key & 7);
        if (r.err) break;
        switch (f) {
        case 1: m->ir_version = (int64_t)rvarint(&r); break;
        case 2: m->producer_name = rbytes(&r); break;
        case 3: m->producer_version = rbytes(&r); break;
        case 7: { rd gs = rsub(&r); if (gs.err || dec_graph(m, gs)) bad = 1; } break;
        case 8: { rd os = rsub(&r);
                  while (os.p < os.end && !os.err) { uint64_t k = rvarint(&os);
                      if ((k >> 3) == 2) m->opset_version = (int64_t)rvarint(&os); else rskip(&os, (int)(k & 7)); }
                  if (os.err) bad = 1; } break;
        default: rskip(&r, wt);
        }
    }
    if (r.err || bad) { orc_model_free(m); return ORC_ERR_PROTO_DECODE; }
    *out = m; return ORC_OK;
}

void orc_model_free(orc_model* m) {
    if (!m) return;
    blk* b = (blk*)m->arena;
    while (b) { blk* nx = b->next; free(b); b = nx; }
    free(m->owned); free(m);
}
