// Row N2: the object level of the on-disk HMSG format written at speed.
// Reference: memory/hmsg/graph/object.py:37-57 (Object.save: <object_id>.ply through o3d.io.write_point_cloud and
// <object_id>.json through json.dump of {object_id, vertices, room_id, name, embedding, view_ids, best_view_id}), driven
// per node by graph.py:1801-1824 save_hmsg_graph.  A 1000-frame scene has ~15 000 objects; their records hold ~3*10^7
// numbers that Python turns into text one float at a time.  Here the instance clouds and pooled features are read back
// from HBM once and a pool of host threads formats and writes the files.
//
// The text is byte-for-byte what json.dump produces: keys in the reference's order, ", " / ": " separators, and every
// number printed like Python's float.__repr__ (shortest digits that round-trip -- std::to_chars gives the same digits --
// laid out with Python's rule: fixed notation while -4 <= exponent < 16, else d.ddde+XX).  Strings come from the caller
// already JSON-encoded (it owns the ids, names and view lists); only numbers are produced here.
#include "hmsg_common.h"

#include <atomic>
#include <charconv>
#include <thread>

namespace {

// Python repr(float) into out (at most 25 bytes), returns the end
char* py_repr(char* out, double v) {
    if (v != v) {
        memcpy(out, "NaN", 3);
        return out + 3;
    }
    if (v == __builtin_inf() || v == -__builtin_inf()) {
        const char* t = v > 0 ? "Infinity" : "-Infinity";
        const size_t n = strlen(t);
        memcpy(out, t, n);
        return out + n;
    }
    char buf[40];
    auto res = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);
    // buf = [-]d[.ddd]e[+-]XX
    char* p = buf;
    if (*p == '-') *out++ = *p++;
    char digits[24];
    int nd = 0;
    while (p < res.ptr && *p != 'e') {
        if (*p != '.') digits[nd++] = *p;
        ++p;
    }
    int e10 = 0;
    {
        ++p;                                   // 'e'
        const bool neg = *p == '-';
        ++p;
        while (p < res.ptr) e10 = e10 * 10 + (*p++ - '0');
        if (neg) e10 = -e10;
    }
    const int decpt = e10 + 1;                 // position of the decimal point relative to the digit string
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) {
            *out++ = '0';
            *out++ = '.';
            for (int i = 0; i < -decpt; ++i) *out++ = '0';
            memcpy(out, digits, (size_t)nd);
            out += nd;
        } else if (decpt >= nd) {
            memcpy(out, digits, (size_t)nd);
            out += nd;
            for (int i = nd; i < decpt; ++i) *out++ = '0';
            *out++ = '.';
            *out++ = '0';
        } else {
            memcpy(out, digits, (size_t)decpt);
            out += decpt;
            *out++ = '.';
            memcpy(out, digits + decpt, (size_t)(nd - decpt));
            out += nd - decpt;
        }
    } else {
        *out++ = digits[0];
        if (nd > 1) {
            *out++ = '.';
            memcpy(out, digits + 1, (size_t)(nd - 1));
            out += nd - 1;
        }
        *out++ = 'e';
        int e = decpt - 1;
        *out++ = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        char eb[8];
        int ne = 0;
        do {
            eb[ne++] = (char)('0' + e % 10);
            e /= 10;
        } while (e);
        if (ne < 2) eb[ne++] = '0';
        while (ne) *out++ = eb[--ne];
    }
    return out;
}

bool write_file(const std::string& path, const char* data, size_t n) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(data, 1, n, f) == n;
    return (fclose(f) == 0) && ok;
}

}  // namespace

extern "C" int hmsg_save_objects(hmsg_t* hc, const char* dir, int64_t n, const hmsg_object_record* recs, int32_t n_threads) {
    hmsg_ctx* h = hc;
    if (!h || !dir || n < 0 || (n > 0 && !recs)) return HMSG_ERR_INVALID;
    try {
        HMSG_REQUIRE((h->merged || h->tree_partial) && h->pooled, HMSG_ERR_INVALID, "hmsg_save_objects: instances are not merged and pooled");
        const int64_t NI = (int64_t)h->inst.off.size() - 1;
        const int D = h->cfg.feat_dim;
        for (int64_t i = 0; i < n; ++i)
            HMSG_REQUIRE(recs[i].instance >= 0 && recs[i].instance < NI && recs[i].file_stem && recs[i].object_id_json &&
                             recs[i].room_id_json && recs[i].name_json && recs[i].view_ids_json && recs[i].best_view_id_json,
                         HMSG_ERR_INVALID, "hmsg_save_objects: bad record");
        if (n == 0) return HMSG_OK;
        HIP_TRY(hipSetDevice(h->cfg.device_id));
        std::vector<double> pts((size_t)std::max<long long>(h->inst.total, 1) * 3);
        std::vector<float> feats((size_t)NI * D);
        if (h->inst.total) HIP_TRY(hipMemcpy(pts.data(), h->inst.pts.p, (size_t)h->inst.total * 24, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(feats.data(), h->inst_feats.p, feats.size() * 4, hipMemcpyDeviceToHost));
        int nt = n_threads > 0 ? n_threads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
        nt = (int)std::min<int64_t>(nt, n);
        std::atomic<int64_t> next{0};
        std::atomic<int> failed{0};
        const std::string base = std::string(dir) + "/";
        auto work = [&]() {
            std::string js;
            std::vector<char> ply;
            for (;;) {
                const int64_t i = next.fetch_add(1);
                if (i >= n || failed.load()) return;
                const hmsg_object_record& r = recs[i];
                const long long a = h->inst.off[(size_t)r.instance], b = h->inst.off[(size_t)r.instance + 1];
                const double* P = pts.data() + a * 3;
                const long long np = b - a;
                // ---- <stem>.ply: binary little-endian, double x y z (a colourless Open3D cloud, Open3D's header comment line)
                char hdr[200];
                const int hl = snprintf(hdr, sizeof(hdr),
                                        "ply\nformat binary_little_endian 1.0\ncomment Created by Open3D\nelement vertex %lld\nproperty double x\n"
                                        "property double y\nproperty double z\nend_header\n", np);
                ply.resize((size_t)hl + (size_t)np * 24);
                memcpy(ply.data(), hdr, (size_t)hl);
                if (np) memcpy(ply.data() + hl, P, (size_t)np * 24);
                if (!write_file(base + r.file_stem + ".ply", ply.data(), ply.size())) {
                    failed.store(1);
                    return;
                }
                // ---- <stem>.json (object.py:46-56 key order)
                js.clear();
                js.reserve((size_t)np * 44 + (size_t)D * 24 + 256);
                js += "{\"object_id\": ";
                js += r.object_id_json;
                js += ", \"vertices\": [";
                char num[32];
                for (long long k = 0; k < np; ++k) {          // vertices = points[:, [0, 2]] (graph.py:1715)
                    if (k) js += ", ";
                    js += '[';
                    js.append(num, (size_t)(py_repr(num, P[k * 3]) - num));
                    js += ", ";
                    js.append(num, (size_t)(py_repr(num, P[k * 3 + 2]) - num));
                    js += ']';
                }
                js += "], \"room_id\": ";
                js += r.room_id_json;
                js += ", \"name\": ";
                js += r.name_json;
                js += ", \"embedding\": [";
                const float* E = feats.data() + (size_t)r.instance * D;
                for (int k = 0; k < D; ++k) {
                    if (k) js += ", ";
                    js.append(num, (size_t)(py_repr(num, (double)E[k]) - num));
                }
                js += "], \"view_ids\": ";
                js += r.view_ids_json;
                js += ", \"best_view_id\": ";
                js += r.best_view_id_json;
                js += '}';
                if (!write_file(base + r.file_stem + ".json", js.data(), js.size())) {
                    failed.store(1);
                    return;
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
        HMSG_REQUIRE(!failed.load(), HMSG_ERR_INVALID, std::string("hmsg_save_objects: cannot write into ") + dir);
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        h->err = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        h->err = e.what();
        return HMSG_ERR_INVALID;
    } catch (...) {
        h->err = "unknown error";
        return HMSG_ERR_INVALID;
    }
}

// ---- the other node records of save_hmsg_graph (graph.py:1801-1824): floors/<f>.{ply,json} (floor.py:37-52),
// rooms/<f>_<r>.{ply,json} (room.py:309-337), views/<id>.json (view.py:56-74).  Their keys differ, their numbers are all printed by
// json.dump the same way: a record is a list of fields in the reference's key order; strings, id lists, null and anything else
// that is not a number array come from the caller JSON-encoded (HMSG_JSON_RAW), numbers are printed here like Python does --
// float64 / float32 arrays and scalars as float.__repr__ of the double, integer arrays as decimal integers.
extern "C" int hmsg_write_json(const char* path, int32_t n_fields, const hmsg_json_field* f) {
    if (!path || n_fields < 0 || (n_fields > 0 && !f)) return HMSG_ERR_INVALID;
    try {
        std::string js = "{";
        char num[40];
        auto put = [&](const hmsg_json_field& fd, int64_t i) {
            if (fd.kind == HMSG_JSON_F64) js.append(num, (size_t)(py_repr(num, ((const double*)fd.data)[i]) - num));
            else if (fd.kind == HMSG_JSON_F32) js.append(num, (size_t)(py_repr(num, (double)((const float*)fd.data)[i]) - num));
            else js += std::to_string((long long)((const int64_t*)fd.data)[i]);
        };
        for (int32_t k = 0; k < n_fields; ++k) {
            const hmsg_json_field& fd = f[k];
            if (!fd.key || fd.ndim < 0 || fd.ndim > 2 || fd.n0 < 0 || fd.n1 < 0) return HMSG_ERR_INVALID;
            if (fd.kind != HMSG_JSON_RAW && fd.kind != HMSG_JSON_F64 && fd.kind != HMSG_JSON_F32 && fd.kind != HMSG_JSON_I64) return HMSG_ERR_INVALID;
            const int64_t cnt = fd.ndim == 0 ? 1 : (fd.ndim == 1 ? fd.n0 : fd.n0 * fd.n1);
            if (!fd.data && (fd.kind == HMSG_JSON_RAW || cnt > 0)) return HMSG_ERR_INVALID;
            if (k) js += ", ";
            js += '"';
            js += fd.key;
            js += "\": ";
            if (fd.kind == HMSG_JSON_RAW) {
                js += (const char*)fd.data;
            } else if (fd.ndim == 0) {
                put(fd, 0);
            } else if (fd.ndim == 1) {
                js += '[';
                for (int64_t i = 0; i < fd.n0; ++i) {
                    if (i) js += ", ";
                    put(fd, i);
                }
                js += ']';
            } else {
                js += '[';
                for (int64_t i = 0; i < fd.n0; ++i) {
                    if (i) js += ", ";
                    js += '[';
                    for (int64_t j = 0; j < fd.n1; ++j) {
                        if (j) js += ", ";
                        put(fd, i * fd.n1 + j);
                    }
                    js += ']';
                }
                js += ']';
            }
        }
        js += '}';
        return write_file(path, js.data(), js.size()) ? HMSG_OK : HMSG_ERR_INVALID;
    } catch (const std::exception&) {
        return HMSG_ERR_INVALID;
    } catch (...) {
        return HMSG_ERR_INVALID;
    }
}

// <stem>.ply as Open3D 0.18's write_point_cloud writes a cloud without colours / normals (the header of hmsg_save_objects)
extern "C" int hmsg_write_ply(const char* path, const double* xyz, int64_t n) {
    if (!path || n < 0 || (n > 0 && !xyz)) return HMSG_ERR_INVALID;
    char hdr[200];
    const int hl = snprintf(hdr, sizeof(hdr),
                            "ply\nformat binary_little_endian 1.0\ncomment Created by Open3D\nelement vertex %lld\nproperty double x\n"
                            "property double y\nproperty double z\nend_header\n", (long long)n);
    try {
        std::vector<char> ply((size_t)hl + (size_t)n * 24);
        memcpy(ply.data(), hdr, (size_t)hl);
        if (n) memcpy(ply.data() + hl, xyz, (size_t)n * 24);
        return write_file(path, ply.data(), ply.size()) ? HMSG_OK : HMSG_ERR_INVALID;
    } catch (const std::exception&) {
        return HMSG_ERR_INVALID;
    } catch (...) {
        return HMSG_ERR_INVALID;
    }
}

// ---- load side: the object table of a saved graph straight into a retrieval index
// (object.py:75-91: metadata["embedding"] -> np.array, float64; graph.py:1892-1987 load_hmsg_graph walks the objects/
//  directory).  The caller lists the objects (file stems, in table order) and their rooms; host threads read
//  <dir>/<stem>.json, find the "embedding" array and parse its numbers with strtod (correctly rounded: the value Python's
//  json module yields); the table goes to hmsg_index_create as float64.
namespace {

// parses a JSON array of numbers starting at p (which points at '['); returns false on anything else
bool parse_number_array(const char* p, const char* end, std::vector<double>& out) {
    if (p >= end || *p != '[') return false;
    ++p;
    for (;;) {
        while (p < end && (*p == ' ' || *p == ',' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
        if (p >= end) return false;
        if (*p == ']') return true;
        if (!strncmp(p, "NaN", 3)) {
            out.push_back(__builtin_nan(""));
            p += 3;
        } else if (!strncmp(p, "Infinity", 8)) {
            out.push_back(__builtin_inf());
            p += 8;
        } else if (!strncmp(p, "-Infinity", 9)) {
            out.push_back(-__builtin_inf());
            p += 9;
        } else {
            char* q = nullptr;
            const double v = strtod(p, &q);
            if (q == p) return false;
            out.push_back(v);
            p = q;
        }
    }
}

}  // namespace

extern "C" int hmsg_index_load_objects(int32_t device_id, const char* dir, int64_t n, const char* const* stems,
                                       const int32_t* room_of_node, int32_t n_threads, hmsg_index_t** out, int32_t* feat_dim) {
    if (!dir || n <= 0 || !stems || !room_of_node || !out) return HMSG_ERR_INVALID;
    *out = nullptr;
    std::vector<std::vector<double>> rows((size_t)n);
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{0};
    const std::string base = std::string(dir) + "/";
    auto work = [&]() {
        std::string text;
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n || failed.load()) return;
            FILE* f = stems[i] ? fopen((base + stems[i] + ".json").c_str(), "rb") : nullptr;
            if (!f) {
                failed.store(1);
                return;
            }
            fseek(f, 0, SEEK_END);
            const long len = ftell(f);
            fseek(f, 0, SEEK_SET);
            text.resize((size_t)std::max<long>(len, 0));
            const bool ok = len >= 0 && fread(&text[0], 1, (size_t)len, f) == (size_t)len;
            fclose(f);
            // (the key cannot occur inside another value: ids, names and view lists are the only strings of a record)
            const size_t k = ok ? text.find("\"embedding\":") : std::string::npos;
            if (k == std::string::npos) {
                failed.store(2);
                return;
            }
            const char* p = text.data() + k + 12;
            const char* end = text.data() + text.size();
            while (p < end && *p == ' ') ++p;
            if (!parse_number_array(p, end, rows[(size_t)i])) {     // "" (an object saved without an embedding) lands here
                failed.store(2);
                return;
            }
        }
    };
    int nt = n_threads > 0 ? n_threads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    nt = (int)std::min<int64_t>(nt, n);
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (failed.load()) {
        fprintf(stderr, "hmsg_index_load_objects: %s under %s\n",
                failed.load() == 1 ? "an object record cannot be opened" : "an object record has no numeric \"embedding\" array", dir);
        return HMSG_ERR_INVALID;
    }
    const size_t D = rows[0].size();
    if (D == 0) return HMSG_ERR_INVALID;
    std::vector<double> table((size_t)n * D);
    for (int64_t i = 0; i < n; ++i) {
        if (rows[(size_t)i].size() != D) {
            fprintf(stderr, "hmsg_index_load_objects: embeddings of different lengths (%zu and %zu)\n", D, rows[(size_t)i].size());
            return HMSG_ERR_INVALID;
        }
        memcpy(&table[(size_t)i * D], rows[(size_t)i].data(), D * 8);
    }
    if (feat_dim) *feat_dim = (int32_t)D;
    return hmsg_index_create(device_id, (int32_t)D, n, table.data(), 1, room_of_node, out);
}

// test hook: Python-repr formatting of doubles, newline separated (tests/test_persist_golden.py compares with repr())
extern "C" int64_t hmsg_test_format_doubles(const double* v, int64_t n, char* out, int64_t cap) {
    int64_t used = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (used + 34 > cap) return -1;
        char* e = py_repr(out + used, v[i]);
        *e++ = '\n';
        used = e - out;
    }
    return used;
}

// test hook (include/hmsg_test.h): DevCache carving
extern "C" int hmsg_test_allocator_carving(int32_t device_id, int32_t root_gb) {
    try {
        HIP_TRY(hipSetDevice(device_id));
        DevCache& c = dev_cache();
        const size_t GB = (size_t)1 << 30, root_bytes = (size_t)root_gb * GB;
        unsigned char* root = nullptr;
        {
            DevBuf<unsigned char> big;
            big.alloc(root_bytes);
            root = big.p;
        }                                                       // parked
        DevBuf<unsigned char> a, b, d;
        {
            CarveScope carve;
            a.alloc(3 * GB);
            b.alloc(GB);
            d.alloc(2 * GB + 12345);
        }
        auto inside = [&](const DevBuf<unsigned char>& x) { return x.p >= root && x.p + x.cap_bytes <= root + root_bytes; };
        if (!inside(a) || !inside(b) || !inside(d)) return 1;
        if (!(a.p + a.cap_bytes <= b.p && b.p + b.cap_bytes <= d.p)) return 2;          // cut off the front, in order
        DevBuf<unsigned char> e;
        e.alloc(3 * GB);                                        // outside a scope: never carved (a block of its own)
        if (inside(e)) return 3;
        b.release();
        c.trim();                                               // pieces out: the root must survive
        if (c.roots_.size() != 1) return 4;
        a.release();
        d.release();
        e.release();
        DevBuf<unsigned char> again;
        again.alloc(root_bytes);                                // every piece is back: the whole block, same address
        if (again.p != root) return 5;
        if (!c.roots_.empty() || !c.piece_root_.empty()) return 6;
        again.release();
        c.trim();
        return 0;
    } catch (const hmsg_error& e) {
        fprintf(stderr, "hmsg_test_allocator_carving: %s\n", e.msg.c_str());
        return -1;
    }
}

// The numbers of one key of a saved record, flattened in file order (load side of floors / rooms: "vertices", "embeddings",
// "clip_embeddings", "represent_images", ...: floor.py:54-67, room.py:339-374 read them with json.load + np.array): strtod is
// correctly rounded, i.e. the float64 Python's json module yields.  *n = numbers in the file (may exceed capacity: call again).
extern "C" int hmsg_read_json_numbers(const char* path, const char* key, double* out, int64_t capacity, int64_t* n) {
    if (!path || !key || !n || capacity < 0 || (capacity > 0 && !out)) return HMSG_ERR_INVALID;
    *n = 0;
    FILE* f = fopen(path, "rb");
    if (!f) return HMSG_ERR_INVALID;
    std::string txt;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) txt.append(buf, got);
    fclose(f);
    const std::string pat = std::string("\"") + key + "\":";
    size_t at = txt.find(pat);
    if (at == std::string::npos) return HMSG_ERR_INVALID;
    const char* p = txt.data() + at + pat.size();
    const char* end = txt.data() + txt.size();
    while (p < end && *p == ' ') ++p;
    if (p >= end) return HMSG_ERR_INVALID;
    int depth = 0;
    int64_t cnt = 0;
    do {                                                   // a scalar, or (nested) arrays of numbers
        while (p < end && (*p == ' ' || *p == ',' || *p == '\n')) ++p;
        if (p >= end) return HMSG_ERR_INVALID;
        if (*p == '[') {
            ++depth;
            ++p;
        } else if (*p == ']') {
            --depth;
            ++p;
        } else {
            double v;
            if (!strncmp(p, "NaN", 3)) { v = __builtin_nan(""); p += 3; }
            else if (!strncmp(p, "Infinity", 8)) { v = __builtin_inf(); p += 8; }
            else if (!strncmp(p, "-Infinity", 9)) { v = -__builtin_inf(); p += 9; }
            else {
                char* q = nullptr;
                v = strtod(p, &q);
                if (q == p) return HMSG_ERR_INVALID;       // (a string, null, an object: not a number array)
                p = q;
            }
            if (cnt < capacity) out[cnt] = v;
            ++cnt;
        }
    } while (depth > 0);
    *n = cnt;
    return HMSG_OK;
}

// ------------------------------------------------------------------------------------------ A9 / A11 bookkeeping (host only)
// utils/graph_utils.py:257-291
extern "C" int hmsg_assign_cameras_to_rooms(const double* dist, int64_t n_cams, int32_t n_rooms, const double* cam_height, double y_min,
                                            double y_max, int32_t* room_of_cam, int64_t* room_off, int32_t* room_imgs) {
    if (n_cams < 0 || n_rooms < 0 || !room_off || (n_cams > 0 && (!cam_height || !room_of_cam)) ||
        (n_cams > 0 && n_rooms > 0 && !dist) || ((n_cams > 0 || n_rooms > 0) && !room_imgs))
        return HMSG_ERR_INVALID;
    try {
        std::vector<std::vector<int32_t>> lists((size_t)n_rooms);
        for (int64_t i = 0; i < n_cams; ++i) {
            const bool inside = !(cam_height[i] < y_min || cam_height[i] > y_max);
            room_of_cam[i] = -1;
            if (!inside || n_rooms == 0) continue;
            int32_t best = 0;
            for (int32_t r = 1; r < n_rooms; ++r)
                if (dist[i * n_rooms + r] < dist[i * n_rooms + best]) best = r;      // (first minimum: np.argmin)
            room_of_cam[i] = best;
            lists[(size_t)best].push_back((int32_t)i);
        }
        for (int32_t r = 0; r < n_rooms; ++r) {
            if (!lists[(size_t)r].empty() || n_cams == 0) continue;
            // closest = np.where(inside, inf, dist[:, r]); np.argmin (camera 0 when every entry is inf)
            int64_t best = 0;
            double bd = HUGE_VAL;
            for (int64_t i = 0; i < n_cams; ++i) {
                const bool inside = !(cam_height[i] < y_min || cam_height[i] > y_max);
                const double d = inside ? HUGE_VAL : dist[i * n_rooms + r];
                if (d < bd) {
                    bd = d;
                    best = i;
                }
            }
            lists[(size_t)r].push_back((int32_t)best);
        }
        int64_t at = 0;
        for (int32_t r = 0; r < n_rooms; ++r) {
            room_off[r] = at;
            for (int32_t v : lists[(size_t)r]) room_imgs[at++] = v;
        }
        room_off[n_rooms] = at;
        return HMSG_OK;
    } catch (const std::bad_alloc&) {
        return HMSG_ERR_NOMEM;
    }
}

// utils/graph_utils.py:334-352
extern "C" int hmsg_pick_representative_views(const float* embs, int64_t n, int32_t dim, const int32_t* labels, const float* centers,
                                              int32_t k, int32_t* out_member, int32_t* n_out) {
    if (!n_out || n < 0 || dim <= 0 || k < 0 || (n > 0 && (!embs || !labels)) || (k > 0 && (!centers || !out_member))) return HMSG_ERR_INVALID;
    *n_out = 0;
    for (int64_t i = 0; i < n; ++i)
        if (labels[i] < 0 || labels[i] >= k) return HMSG_ERR_INVALID;
    for (int32_t lab = 0; lab < k; ++lab) {
        int64_t best = -1;
        double bs = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            if (labels[i] != lab) continue;
            double sc = 0.0;
            for (int32_t d = 0; d < dim; ++d) sc += (double)embs[i * dim + d] * (double)centers[(int64_t)lab * dim + d];
            if (best < 0 || sc > bs) {
                best = i;
                bs = sc;
            }
        }
        if (best >= 0) out_member[(*n_out)++] = (int32_t)best;
    }
    return HMSG_OK;
}

// graph.py:1752-1775
extern "C" int hmsg_graph_edges(int32_t n_floors, int32_t n_rooms, const int32_t* room_floor, int32_t n_objects, const int32_t* obj_room,
                                int32_t n_views, const int32_t* view_room, const int64_t* view_obj_off, const int32_t* view_obj,
                                int64_t* edges, int64_t capacity, int64_t* n_edges) {
    if (!n_edges || n_floors < 0 || n_rooms < 0 || n_objects < 0 || n_views < 0 || capacity < 0 || (n_rooms > 0 && !room_floor) ||
        (n_objects > 0 && !obj_room) || (n_views > 0 && (!view_room || !view_obj_off)) || (capacity > 0 && !edges))
        return HMSG_ERR_INVALID;
    *n_edges = 0;
    for (int32_t r = 0; r < n_rooms; ++r)
        if (room_floor[r] < 0 || room_floor[r] >= n_floors) return HMSG_ERR_INVALID;
    for (int32_t o = 0; o < n_objects; ++o)
        if (obj_room[o] < -1 || obj_room[o] >= n_rooms) return HMSG_ERR_INVALID;
    try {
        const int64_t f0 = 1, r0 = f0 + n_floors, o0 = r0 + n_rooms, v0 = o0 + n_objects;
        std::vector<std::vector<int32_t>> rooms_of((size_t)n_floors), objs_of((size_t)n_rooms);
        for (int32_t r = 0; r < n_rooms; ++r) rooms_of[(size_t)room_floor[r]].push_back(r);
        for (int32_t o = 0; o < n_objects; ++o)
            if (obj_room[o] >= 0) objs_of[(size_t)obj_room[o]].push_back(o);
        int64_t cnt = 0;
        auto emit = [&](int64_t a, int64_t b) {
            if (cnt < capacity) {
                edges[cnt * 2] = a;
                edges[cnt * 2 + 1] = b;
            }
            ++cnt;
        };
        for (int32_t f = 0; f < n_floors; ++f) {
            emit(0, f0 + f);
            for (int32_t r : rooms_of[(size_t)f]) {
                emit(f0 + f, r0 + r);
                for (int32_t o : objs_of[(size_t)r]) emit(r0 + r, o0 + o);
            }
        }
        std::vector<int32_t> objs;
        for (int32_t v = 0; v < n_views; ++v) {
            if (view_room[v] >= n_rooms) return HMSG_ERR_INVALID;
            if (view_room[v] >= 0) emit(r0 + view_room[v], v0 + v);
            const int64_t a = view_obj_off[v], b = view_obj_off[v + 1];
            if (a > b || (b > a && !view_obj)) return HMSG_ERR_INVALID;
            objs.assign(view_obj + a, view_obj + b);
            std::sort(objs.begin(), objs.end());
            objs.erase(std::unique(objs.begin(), objs.end()), objs.end());
            for (int32_t o : objs) {
                if (o < 0 || o >= n_objects) return HMSG_ERR_INVALID;
                emit(v0 + v, o0 + o);
            }
        }
        *n_edges = cnt;
        return cnt <= capacity ? HMSG_OK : HMSG_ERR_INVALID;
    } catch (const std::bad_alloc&) {
        return HMSG_ERR_NOMEM;
    }
}
