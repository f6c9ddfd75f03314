// loader.cpp — safetensors ingestion for the DiT and the VAE (replaces the file side of
// Loading/WeightLoader.swift:24-78 `loadWeights(from:)` + the key handling of :306-345).
//
// Format: 8-byte little-endian header length N, N bytes of JSON
//   { "<key>": {"dtype": "BF16"|"F16"|"F32"|"I8", "shape": [...], "data_offsets": [begin, end]}, ...,
//     "__metadata__": {...} }
// followed by the tensor bytes.  The file is mmap'ed and every tensor the model's registry knows is
// handed to flux2mi_{dit,vae}_set_tensor straight from the mapping (the H2D copy is the only copy).
// Keys: an optional "transformer." / "vae." prefix is dropped (WeightLoader.swift:322-325); quanto
// int8 pairs "<key>._data" + "<key>._scale" are de-quantised on the device; "*.input_scale" /
// "*.output_scale" are skipped (:336-338); other unknown keys are counted and skipped.
#include <errno.h>
#include <stdlib.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "loader.h"
#include "../../include/flux2mi.h"

namespace {

// Minimal JSON reader for the safetensors header (objects, arrays, strings, integers).
struct Json {
    const char* p;
    const char* e;
    bool ok = true;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool eat(char c) { ws(); if (p < e && *p == c) { ++p; return true; } return false; }
    std::string str() {
        std::string s;
        ws();
        if (p >= e || *p != '"') { ok = false; return s; }
        ++p;
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) {
                ++p;
                switch (*p) {
                    case 'n': s += '\n'; break;
                    case 't': s += '\t'; break;
                    case 'u': s += '?'; p += 4; break;   // keys of interest are ASCII
                    default: s += *p;
                }
                ++p;
            } else {
                s += *p++;
            }
        }
        if (p >= e) { ok = false; return s; }
        ++p;
        return s;
    }
    // non-negative integer < 2^62 (shapes and byte offsets): a sign or an overflowing literal is a malformed header
    int64_t integer() {
        ws();
        if (p >= e || *p < '0' || *p > '9') { ok = false; return 0; }
        int64_t v = 0;
        while (p < e && *p >= '0' && *p <= '9') {
            if (v > ((int64_t)1 << 62) / 10 - 1) { ok = false; return 0; }
            v = v * 10 + (*p++ - '0');
        }
        return v;
    }
    void skip_value() {   // any JSON value
        ws();
        if (p >= e) { ok = false; return; }
        if (*p == '"') { (void)str(); return; }
        if (*p == '{' || *p == '[') {
            const char open = *p, close = (*p == '{') ? '}' : ']';
            int depth = 0;
            while (p < e) {
                if (*p == '"') { (void)str(); continue; }
                if (*p == open) ++depth;
                else if (*p == close && --depth == 0) { ++p; return; }
                ++p;
            }
            ok = false;
            return;
        }
        while (p < e && *p != ',' && *p != '}' && *p != ']') ++p;
    }
};

}  // namespace

MappedFile::~MappedFile() {
    if (base) munmap(base, size);
    if (fd >= 0) close(fd);
}

int parse_safetensors(const char* path, MappedFile& f, std::map<std::string, TensorInfo>& out, const uint8_t** data,
                      std::map<std::string, std::string>* metadata, uint64_t* header_len) {
    f.fd = open(path, O_RDONLY);
    if (f.fd < 0) {
        flux2mi_set_error("load_safetensors: cannot open '%s': %s", path, strerror(errno));
        return FLUX2MI_ERR_INVALID_ARG;
    }
    struct stat st;
    if (fstat(f.fd, &st) != 0 || st.st_size < 8) {
        flux2mi_set_error("load_safetensors: '%s' is not a safetensors file (size)", path);
        return FLUX2MI_ERR_INVALID_ARG;
    }
    f.size = (size_t)st.st_size;
    void* m = mmap(nullptr, f.size, PROT_READ, MAP_PRIVATE, f.fd, 0);
    if (m == MAP_FAILED) {
        flux2mi_set_error("load_safetensors: mmap('%s') failed: %s", path, strerror(errno));
        return FLUX2MI_ERR_OOM;
    }
    f.base = static_cast<uint8_t*>(m);
    uint64_t hlen = 0;
    memcpy(&hlen, f.base, 8);
    if (hlen == 0 || hlen > f.size - 8) {
        flux2mi_set_error("load_safetensors: '%s': bad header length %llu", path, (unsigned long long)hlen);
        return FLUX2MI_ERR_INVALID_ARG;
    }
    Json j{reinterpret_cast<const char*>(f.base + 8), reinterpret_cast<const char*>(f.base + 8 + hlen)};
    if (!j.eat('{')) j.ok = false;
    while (j.ok) {
        if (j.eat('}')) break;
        std::string key = j.str();
        if (!j.ok || !j.eat(':')) { j.ok = false; break; }
        if (key == "__metadata__") {
            if (!metadata) {
                j.skip_value();
            } else {   // flat string -> string object
                if (!j.eat('{')) { j.ok = false; break; }
                while (j.ok) {
                    if (j.eat('}')) break;
                    std::string k = j.str();
                    if (!j.eat(':')) { j.ok = false; break; }
                    j.ws();
                    if (j.p < j.e && *j.p == '"') (*metadata)[k] = j.str();
                    else j.skip_value();
                    (void)j.eat(',');
                }
            }
        } else {
            TensorInfo t;
            if (!j.eat('{')) { j.ok = false; break; }
            while (j.ok) {
                if (j.eat('}')) break;
                std::string field = j.str();
                if (!j.eat(':')) { j.ok = false; break; }
                if (field == "dtype") t.dtype = j.str();
                else if (field == "shape") {
                    if (!j.eat('[')) { j.ok = false; break; }
                    while (j.ok && !j.eat(']')) { t.shape.push_back(j.integer()); (void)j.eat(','); }
                } else if (field == "data_offsets") {
                    if (!j.eat('[')) { j.ok = false; break; }
                    t.begin = (uint64_t)j.integer(); (void)j.eat(',');
                    t.end = (uint64_t)j.integer();
                    if (!j.eat(']')) j.ok = false;
                } else j.skip_value();
                (void)j.eat(',');
            }
            // payload = f.size - 8 - hlen bytes (hlen <= f.size - 8 checked above): no sum that can wrap
            if (!j.ok) break;
            if (t.begin > t.end || t.end > (uint64_t)f.size - 8 - hlen) {
                flux2mi_set_error("load_safetensors: '%s': tensor '%s' lies outside the file", path, key.c_str());
                return FLUX2MI_ERR_INVALID_ARG;
            }
            out[key] = t;
        }
        (void)j.eat(',');
    }
    if (!j.ok) {
        flux2mi_set_error("load_safetensors: '%s': malformed header", path);
        return FLUX2MI_ERR_INVALID_ARG;
    }
    *data = f.base + 8 + hlen;
    if (header_len) *header_len = hlen;
    return FLUX2MI_OK;
}

namespace {

int st_dtype(const std::string& s) {
    if (s == "F32") return FLUX2MI_F32;
    if (s == "BF16") return FLUX2MI_BF16;
    if (s == "F16") return FLUX2MI_F16;
    return -1;
}

// prod(shape) * elem_size == end - begin, without overflow (every dim < 2^62 from the parser)
bool bytes_match(const TensorInfo& t, uint64_t elem_size) {
    uint64_t n = 1;
    for (int64_t d : t.shape) {
        if (d < 0) return false;
        if (d != 0 && n > (UINT64_MAX / elem_size) / (uint64_t)d) return false;
        n *= (uint64_t)d;
    }
    return n * elem_size == t.end - t.begin;
}

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

template <class SetFn, class Q8Fn>
int load_all(const char* path, const char* const* prefixes, SetFn set_tensor, Q8Fn set_q8, int* n_loaded, int* n_skipped) {
    MappedFile f;
    std::map<std::string, TensorInfo> ts;
    const uint8_t* data = nullptr;
    F2_TRY(parse_safetensors(path, f, ts, &data));
    int loaded = 0, skipped = 0;
    for (const auto& kv : ts) {
        std::string key = kv.first;
        for (const char* const* pf = prefixes; *pf; ++pf)
            if (key.compare(0, strlen(*pf), *pf) == 0) { key = key.substr(strlen(*pf)); break; }
        const TensorInfo& t = kv.second;
        if (ends_with(key, ".input_scale") || ends_with(key, ".output_scale") || ends_with(key, "._scale")) continue;
        if (ends_with(key, "._data")) {   // quanto qint8: needs its "._scale" partner
            const std::string base = key.substr(0, key.size() - 6);
            auto sc = ts.find(kv.first.substr(0, kv.first.size() - 6) + "._scale");
            if (t.dtype != "I8" || t.shape.size() != 2 || !bytes_match(t, 1) || sc == ts.end() || sc->second.dtype != "F32" ||
                !bytes_match(sc->second, 4) || sc->second.end - sc->second.begin != (uint64_t)t.shape[0] * 4) {
                flux2mi_set_error("load_safetensors: '%s': quantised tensor '%s' without a matching f32 per-row ._scale",
                                  path, kv.first.c_str());
                return FLUX2MI_ERR_SHAPE;
            }
            const int r = set_q8(base.c_str(), reinterpret_cast<const int8_t*>(data + t.begin),
                                 reinterpret_cast<const float*>(data + sc->second.begin), t.shape[0], t.shape[1]);
            if (r == FLUX2MI_ERR_UNKNOWN_KEY) { ++skipped; continue; }
            F2_TRY(r);
            ++loaded;
            continue;
        }
        const int dt = st_dtype(t.dtype);
        if (dt < 0) { ++skipped; continue; }
        if (!bytes_match(t, dt == FLUX2MI_F32 ? 4 : 2)) {
            flux2mi_set_error("load_safetensors: '%s': tensor '%s': data_offsets span %llu bytes, shape x dtype says otherwise",
                              path, kv.first.c_str(), (unsigned long long)(t.end - t.begin));
            return FLUX2MI_ERR_SHAPE;
        }
        const int r = set_tensor(key.c_str(), dt, t.shape.data(), (int)t.shape.size(), data + t.begin);
        if (r == FLUX2MI_ERR_UNKNOWN_KEY) { ++skipped; continue; }       // key of another component
        F2_TRY(r);
        ++loaded;
    }
    if (n_loaded) *n_loaded = loaded;
    if (n_skipped) *n_skipped = skipped;
    return FLUX2MI_OK;
}

}  // namespace

extern "C" int flux2mi_dit_load_safetensors(flux2mi_dit* dit, const char* path, int* n_loaded, int* n_skipped) {
    F2_REQUIRE(dit && path, FLUX2MI_ERR_INVALID_ARG, "dit_load_safetensors: NULL argument");
    static const char* const prefixes[] = {"transformer.", "model.diffusion_model.", nullptr};
    return load_all(
        path, prefixes,
        [&](const char* k, int dt, const int64_t* sh, int nd, const void* p) { return flux2mi_dit_set_tensor(dit, k, dt, sh, nd, p); },
        [&](const char* k, const int8_t* d, const float* s, int64_t r, int64_t c) { return flux2mi_dit_set_tensor_q8(dit, k, d, s, r, c); },
        n_loaded, n_skipped);
}

extern "C" int flux2mi_vae_load_safetensors(flux2mi_vae* vae, const char* path, int* n_loaded, int* n_skipped) {
    F2_REQUIRE(vae && path, FLUX2MI_ERR_INVALID_ARG, "vae_load_safetensors: NULL argument");
    static const char* const prefixes[] = {"vae.", nullptr};
    return load_all(
        path, prefixes,
        [&](const char* k, int dt, const int64_t* sh, int nd, const void* p) { return flux2mi_vae_set_tensor(vae, k, dt, sh, nd, p); },
        [&](const char*, const int8_t*, const float*, int64_t, int64_t) { return (int)FLUX2MI_ERR_UNKNOWN_KEY; },
        n_loaded, n_skipped);
}

// Text-encoder checkpoint shards (HF keys; `language_model.` is the prefix of the text model inside the Mistral-Small-3.2 VLM
// checkpoint).  Keys of layers that are not run, lm_head and vision towers are skipped (FLUX2MI_ERR_UNKNOWN_KEY from set_tensor).
extern "C" int flux2mi_textenc_load_safetensors(flux2mi_textenc* te, const char* path, int* n_loaded, int* n_skipped) {
    F2_REQUIRE(te && path, FLUX2MI_ERR_INVALID_ARG, "textenc_load_safetensors: NULL argument");
    static const char* const prefixes[] = {"language_model.", "text_encoder.", nullptr};
    return load_all(
        path, prefixes,
        [&](const char* k, int dt, const int64_t* sh, int nd, const void* p) { return flux2mi_textenc_set_tensor(te, k, dt, sh, nd, p); },
        [&](const char*, const int8_t*, const float*, int64_t, int64_t) { return (int)FLUX2MI_ERR_UNKNOWN_KEY; },
        n_loaded, n_skipped);
}

// `loadLoRA(LoRAConfig)` (Flux2Pipeline.swift:618-663) for the weights side: LoRA/LoRALoader.swift + WeightLoader.mergeLoRAWeights
// (Loading/WeightLoader.swift:736-856).  Reads `<layer>.lora_A.weight` [rank, in] / `<layer>.lora_B.weight` [out, rank] pairs, strips
// the prefixes `base_model.model.`, `transformer.`, `diffusion_model.` one after the other (LoRALoader.swift:132-140), scale =
// (metadata lora_alpha / lora_rank, default 1) x user_scale, and merges W += s B A into the resident bf16 weights.  Layers this
// transformer does not have are skipped like in the reference (:783-789) and counted in *n_unmatched; a file in which NO layer
// matches is an error instead of a LoRA that silently does nothing.
extern "C" int flux2mi_dit_load_lora_safetensors(flux2mi_dit* dit, const char* path, float user_scale, int* n_merged, int* n_unmatched) {
    F2_REQUIRE(dit && path, FLUX2MI_ERR_INVALID_ARG, "dit_load_lora_safetensors: NULL argument");
    MappedFile f;
    std::map<std::string, TensorInfo> ts;
    std::map<std::string, std::string> md;
    const uint8_t* data = nullptr;
    F2_TRY(parse_safetensors(path, f, ts, &data, &md));
    float s = 1.0f;
    {
        auto r = md.find("lora_rank"), a = md.find("lora_alpha");
        if (r != md.end() && a != md.end()) {
            char *e1 = nullptr, *e2 = nullptr;
            const double rank = strtod(r->second.c_str(), &e1), alpha = strtod(a->second.c_str(), &e2);
            if (e1 != r->second.c_str() && e2 != a->second.c_str() && rank > 0) s = (float)(alpha / rank);
        }
    }
    s *= user_scale;
    static const char* const kSuffixA = ".lora_A.weight";
    int merged = 0, unmatched = 0;
    std::string first_unmatched;
    for (const auto& kv : ts) {
        if (!ends_with(kv.first, kSuffixA)) continue;
        const std::string base = kv.first.substr(0, kv.first.size() - strlen(kSuffixA));
        auto b = ts.find(base + ".lora_B.weight");
        if (b == ts.end()) continue;
        const TensorInfo &A = kv.second, &B = b->second;
        const int dta = st_dtype(A.dtype), dtb = st_dtype(B.dtype);
        if (dta < 0 || dtb < 0 || A.shape.size() != 2 || B.shape.size() != 2 || A.shape[0] != B.shape[1] ||
            !bytes_match(A, dta == FLUX2MI_F32 ? 4 : 2) || !bytes_match(B, dtb == FLUX2MI_F32 ? 4 : 2)) {
            flux2mi_set_error("load_lora: '%s': pair '%s': float A [rank, in] / B [out, rank] expected", path, base.c_str());
            return FLUX2MI_ERR_SHAPE;
        }
        std::string layer = base;
        for (const char* pre : {"base_model.model.", "transformer.", "diffusion_model."})
            if (layer.compare(0, strlen(pre), pre) == 0) layer = layer.substr(strlen(pre));
        // shapes from the FILE: a LoRA of another model width with the same key names is a shape error, not an over-read of the mapping
        const int r = flux2mi_dit_merge_lora_checked(dit, (layer + ".weight").c_str(), dta, data + A.begin, A.shape[0], A.shape[1], dtb,
                                                     data + B.begin, B.shape[0], B.shape[1], s);
        if (r == FLUX2MI_ERR_UNKNOWN_KEY) {
            if (!unmatched++) first_unmatched = layer;
            continue;
        }
        F2_TRY(r);
        ++merged;
    }
    if (n_merged) *n_merged = merged;
    if (n_unmatched) *n_unmatched = unmatched;
    F2_REQUIRE(merged > 0, FLUX2MI_ERR_INVALID_ARG, "LoRA '%s': none of its %d layers matches this transformer (first: %s)", path, unmatched,
               first_unmatched.c_str());
    return FLUX2MI_OK;
}
