// onnx_pb.cpp — proto3 wire decoder for the ONNX subset + typed tensor views (see onnx_pb.hpp).
// A cursor-based reader with one templated "for each field" walker; repeated scalars are accepted packed or
// unpacked and unknown fields are skipped by wire type, as SwiftProtobuf does for the reference
// (Package.swift:18-22; generated `default: break` arms, e.g. onnx.pb.swift:1158).
#include "onnx_pb.hpp"

#include <string.h>

#include "../../../include/smelter_hip.h"
#include "half.hpp"

namespace smelter {
namespace pb {
namespace {

struct Cursor {
    const uint8_t* p; const uint8_t* end; bool ok = true;
    Cursor(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}
    bool more() const { return ok && p < end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64 && p < end; shift += 7) {
            uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return v;
        }
        ok = false; return 0;
    }
    Cursor sub() {
        uint64_t n = varint();
        if (!ok || n > (uint64_t)(end - p)) { ok = false; return Cursor(p, p); }
        Cursor c(p, p + n); p += n; return c;
    }
    std::string_view bytes() { Cursor c = sub(); return std::string_view((const char*)c.p, (size_t)(c.end - c.p)); }
    template <class T> T fixed() { T v{}; if ((size_t)(end - p) < sizeof(T)) { ok = false; return v; } memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
    void skip(int wt) {
        switch (wt) {
        case 0: varint(); break;
        case 1: fixed<uint64_t>(); break;
        case 2: sub(); break;
        case 5: fixed<uint32_t>(); break;
        default: ok = false;
        }
    }
};

// walk(c, fn): fn(field, wire_type, cursor) must consume the field's payload or return false to have it skipped
template <class F> bool walk(Cursor c, F&& fn) {
    while (c.more()) {
        uint64_t key = c.varint(); if (!c.ok) break;
        int field = (int)(key >> 3), wt = (int)(key & 7);
        if (!fn(field, wt, c)) c.skip(wt);
    }
    return c.ok;
}

template <class T> void rep_varint(Cursor& c, int wt, std::vector<T>* v) {
    if (wt == 2) { Cursor s = c.sub(); while (s.more()) v->push_back((T)s.varint()); if (!s.ok) c.ok = false; }
    else if (wt == 0) v->push_back((T)c.varint());
    else c.ok = false;
}
template <class T> void rep_fixed(Cursor& c, int wt, std::vector<T>* v) {
    if (wt == 2) { Cursor s = c.sub(); size_t n = (size_t)(s.end - s.p) / sizeof(T); size_t o = v->size(); v->resize(o + n); if (n) memcpy(v->data() + o, s.p, n * sizeof(T)); }
    else if ((wt == 5 && sizeof(T) == 4) || (wt == 1 && sizeof(T) == 8)) v->push_back(c.fixed<T>());
    else c.ok = false;
}

bool dec_tensor(Cursor c, Tensor* t) {
    return walk(c, [&](int f, int wt, Cursor& r) {
        switch (f) {
        case 1: rep_varint(r, wt, &t->dims); return true;
        case 2: t->data_type = (int32_t)r.varint(); return true;
        case 4: rep_fixed(r, wt, &t->float_data); return true;
        case 5: rep_varint(r, wt, &t->int32_data); return true;
        case 7: rep_varint(r, wt, &t->int64_data); return true;
        case 8: t->name = r.bytes(); return true;
        case 9: t->raw_data = r.bytes(); return true;
        case 10: rep_fixed(r, wt, &t->double_data); return true;
        case 11: rep_varint(r, wt, &t->uint64_data); return true;
        default: return false;
        }
    });
}

bool dec_attr(Cursor c, Attribute* a) {
    return walk(c, [&](int f, int wt, Cursor& r) {
        switch (f) {
        case 1: a->name = r.bytes(); return true;
        case 2: a->f = r.fixed<float>(); return true;
        case 3: a->i = (int64_t)r.varint(); return true;
        case 4: a->s = r.bytes(); return true;
        case 5: a->has_t = true; if (!dec_tensor(r.sub(), &a->t)) r.ok = false; return true;
        case 7: rep_fixed(r, wt, &a->floats); return true;
        case 8: rep_varint(r, wt, &a->ints); return true;
        case 20: a->type = (int32_t)r.varint(); return true;
        default: return false;
        }
    });
}

bool dec_node(Cursor c, Node* n) {
    return walk(c, [&](int f, int, Cursor& r) {
        switch (f) {
        case 1: n->input.push_back(r.bytes()); return true;
        case 2: n->output.push_back(r.bytes()); return true;
        case 3: n->name = r.bytes(); return true;
        case 4: n->op_type = r.bytes(); return true;
        case 5: n->attribute.emplace_back(); if (!dec_attr(r.sub(), &n->attribute.back())) r.ok = false; return true;
        case 7: n->domain = r.bytes(); return true;
        default: return false;
        }
    });
}

bool dec_value_info(Cursor c, ValueInfo* v) {
    return walk(c, [&](int f, int, Cursor& r) {
        if (f == 1) { v->name = r.bytes(); return true; }
        if (f != 2) return false;
        bool ok = walk(r.sub(), [&](int f2, int, Cursor& ty) {                 // TypeProto
            if (f2 != 1) return false;
            bool ok2 = walk(ty.sub(), [&](int f3, int, Cursor& tt) {           // TypeProto.Tensor
                if (f3 == 1) { v->elem_type = (int32_t)tt.varint(); return true; }
                if (f3 != 2) return false;
                bool ok3 = walk(tt.sub(), [&](int f4, int, Cursor& sh) {       // TensorShapeProto
                    if (f4 != 1) return false;
                    int64_t dv = 0;                                             // dim_param (symbolic) reads as 0
                    bool ok4 = walk(sh.sub(), [&](int f5, int, Cursor& dm) { if (f5 == 1) { dv = (int64_t)dm.varint(); return true; } return false; });
                    if (!ok4) sh.ok = false;
                    v->dims.push_back(dv); return true;
                });
                if (!ok3) tt.ok = false; return true;
            });
            if (!ok2) ty.ok = false; return true;
        });
        if (!ok) r.ok = false;
        return true;
    });
}

bool dec_graph(Cursor c, Graph* g) {
    return walk(c, [&](int f, int, Cursor& r) {
        switch (f) {
        case 1: g->node.emplace_back(); if (!dec_node(r.sub(), &g->node.back())) r.ok = false; return true;
        case 2: g->name = r.bytes(); return true;
        case 5: g->initializer.emplace_back(); if (!dec_tensor(r.sub(), &g->initializer.back())) r.ok = false; return true;
        case 11: g->input.emplace_back(); if (!dec_value_info(r.sub(), &g->input.back())) r.ok = false; return true;
        case 12: g->output.emplace_back(); if (!dec_value_info(r.sub(), &g->output.back())) r.ok = false; return true;
        case 13: g->value_info.emplace_back(); if (!dec_value_info(r.sub(), &g->value_info.back())) r.ok = false; return true;
        default: return false;
        }
    });
}

}  // namespace

bool decode_model(const uint8_t* p, size_t n, Model* m) {
    return walk(Cursor(p, p + n), [&](int f, int, Cursor& r) {
        switch (f) {
        case 1: m->ir_version = (int64_t)r.varint(); return true;
        case 2: m->producer_name = r.bytes(); return true;
        case 3: m->producer_version = r.bytes(); return true;
        case 7: if (!dec_graph(r.sub(), &m->graph)) r.ok = false; return true;
        case 8: { bool ok = walk(r.sub(), [&](int f2, int, Cursor& o) { if (f2 == 2) { m->opset_version = (int64_t)o.varint(); return true; } return false; });
                  if (!ok) r.ok = false; return true; }
        default: return false;
        }
    });
}

// ---- Onnx_TensorProto+Extensions.swift ---------------------------------------------------------------------
namespace {
template <class T> std::vector<T> raw_as(std::string_view raw) {   // Data+Extensions.swift:4-14: count = bytes / stride
    std::vector<T> v(raw.size() / sizeof(T));
    if (!v.empty()) memcpy(v.data(), raw.data(), v.size() * sizeof(T));
    return v;
}
template <class D, class S> void convert(const std::vector<S>& s, std::vector<D>* d) { d->resize(s.size()); for (size_t i = 0; i < s.size(); i++) (*d)[i] = (D)s[i]; }
// Int(Float) / Int(Double) (Onnx_TensorProto+Extensions.swift:21-30) truncate toward zero and TRAP on NaN / out-of-range values; a C++
// cast of such a value is undefined behaviour.  The file is untrusted input: saturate instead (NaN -> 0), never trap.
template <class S> void convert_trunc(const std::vector<S>& s, std::vector<int64_t>* d) {
    d->resize(s.size());
    for (size_t i = 0; i < s.size(); i++) {
        const double v = (double)s[i];
        (*d)[i] = v != v ? 0 : v >= 9.2233720368547748e18 ? INT64_MAX : v <= -9.2233720368547758e18 ? INT64_MIN : (int64_t)v;
    }
}
void half_raw(std::string_view raw, std::vector<float>* out) {      // Data+Extensions.swift:16-24
    size_t n = raw.size() / 2; out->resize(n);
    for (size_t i = 0; i < n; i++) { uint16_t h; memcpy(&h, raw.data() + 2 * i, 2); (*out)[i] = half_bits_to_float(h); }
}
}  // namespace

int tensor_floats(const Tensor& t, std::vector<float>* out) {
    switch (t.data_type) {
    case INT32: case INT16: case INT8: case UINT16: case UINT8: case BOOL: convert(t.int32_data, out); return SM_OK;
    case INT64: convert(t.int64_data, out); return SM_OK;                                   // :46-47 (typed field only)
    case UINT32: case UINT64: convert(t.uint64_data, out); return SM_OK;
    case FLOAT: if (t.float_data.empty()) *out = raw_as<float>(t.raw_data); else *out = t.float_data; return SM_OK;  // :50-54
    case DOUBLE: convert(t.double_data, out); return SM_OK;
    case FLOAT16: half_raw(t.raw_data, out); return SM_OK;                                  // :57-58
    default: return SM_ERR_UNSUPPORTED_DTYPE;                                               // :60 fatalError
    }
}

int tensor_integers(const Tensor& t, std::vector<int64_t>* out) {
    switch (t.data_type) {
    case INT32: case INT16: case INT8: case UINT16: case UINT8: case BOOL: convert(t.int32_data, out); return SM_OK;
    case INT64: if (t.int64_data.empty()) *out = raw_as<int64_t>(t.raw_data); else *out = t.int64_data; return SM_OK;   // :11-17
    case UINT32: case UINT64: convert(t.uint64_data, out); return SM_OK;
    case FLOAT: { std::vector<float> f = t.float_data.empty() ? raw_as<float>(t.raw_data) : t.float_data; convert_trunc(f, out); return SM_OK; }  // :21-27, Int(Float) truncates
    case DOUBLE: convert_trunc(t.double_data, out); return SM_OK;
    case FLOAT16: { std::vector<float> f; half_raw(t.raw_data, &f); convert_trunc(f, out); return SM_OK; }              // :29-30
    default: return SM_ERR_UNSUPPORTED_DTYPE;                                               // :32 fatalError
    }
}

}  // namespace pb
}  // namespace smelter
