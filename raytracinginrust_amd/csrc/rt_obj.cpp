// csrc/rt_obj.cpp — host-side asset ingest: Wavefront OBJ -> (positions, triangle indices), the job
// `tobj::load_obj(path, &tobj::OFFLINE_RENDERING_LOAD_OPTIONS)` does for Mesh::load_obj (src/mesh.rs:33-61; crate tobj 3.2.3,
// not vendored under /root/reference).  What the path uses of it, restated from tobj's documented behaviour:
//   * `v x y z` positions are parsed as f32 (widened to f64 at src/mesh.rs:51, then `* scale + offset`);
//   * `f` elements are `v`, `v/vt`, `v//vn` or `v/vt/vn`; only the position index is used; indices are 1-based, negative ones
//     count back from the positions read so far; polygons are fan-triangulated (triangulate = true);
//   * only models[0] is used (src/mesh.rs:42): a new `o` / `g` after the first faces ends it;
//   * a malformed number, a zero index or an index outside the positions read is an error (tobj returns Err, the reference's
//     caller `.unwrap()`s it, src/main.rs:431) — an error string here, never a panic or an out-of-range access.
// Off the hot path: runs once at scene build.  Plain C++ (no HIP): also built with -fsanitize=address,undefined (make asan).
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rt {

namespace {

const size_t MAX_VERTICES = 1u << 26, MAX_INDICES = 3u << 26;    // far above any real mesh; bounds the allocations a hostile file can cause

// one whitespace-separated token of [p, end): returns false at end of line
bool next_token(const char*& p, const char* end, const char*& tb, const char*& te) {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
    if (p >= end) return false;
    tb = p;
    while (p < end && *p != ' ' && *p != '\t' && *p != '\r') p++;
    te = p;
    return true;
}
bool parse_f32(const char* tb, const char* te, double& out) {
    char buf[64];
    size_t n = (size_t)(te - tb);
    if (n == 0 || n >= sizeof(buf)) return false;
    std::memcpy(buf, tb, n); buf[n] = 0;
    char* endp = nullptr;
    errno = 0;
    float f = std::strtof(buf, &endp);
    if (endp != buf + n) return false;
    out = (double)f;                                               // `p[0] as f64`, src/mesh.rs:51
    return true;
}
// the position index of a face element `v[/vt][/vn]`
bool parse_face_index(const char* tb, const char* te, long long& out) {
    const char* slash = tb;
    while (slash < te && *slash != '/') slash++;
    char buf[32];
    size_t n = (size_t)(slash - tb);
    if (n == 0 || n >= sizeof(buf)) return false;
    std::memcpy(buf, tb, n); buf[n] = 0;
    char* endp = nullptr;
    errno = 0;
    long long v = std::strtoll(buf, &endp, 10);
    if (endp != buf + n || errno == ERANGE) return false;
    out = v;
    return true;
}

} // namespace

bool parse_obj(const char* data, size_t size, const double offset[3], double scale, std::vector<double>& positions,
               std::vector<uint32_t>& indices, std::string& err) {
    positions.clear(); indices.clear();
    bool have_faces = false;
    size_t line_no = 0;
    const char* p = data; const char* const end = data + size;
    auto fail = [&](const std::string& m) { err = m + " (line " + std::to_string(line_no) + ")"; return false; };
    std::vector<uint32_t> vs;
    while (p < end) {
        const char* eol = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        line_no++;
        const char* q = p; const char *tb, *te;
        if (next_token(q, eol, tb, te)) {
            const size_t tl = (size_t)(te - tb);
            if (tl == 1 && tb[0] == 'v') {
                double c[3];
                for (int k = 0; k < 3; k++) {
                    if (!next_token(q, eol, tb, te) || !parse_f32(tb, te, c[k])) return fail("position parse error");
                }
                if (positions.size() / 3 >= MAX_VERTICES) return fail("too many vertices");
                for (int k = 0; k < 3; k++) positions.push_back(c[k] * scale + offset[k]);                 // src/mesh.rs:51
            } else if (tl == 1 && tb[0] == 'f') {
                have_faces = true;
                vs.clear();
                const long long n_pos = (long long)(positions.size() / 3);
                while (next_token(q, eol, tb, te)) {
                    long long k;
                    if (!parse_face_index(tb, te, k)) return fail("face parse error");
                    const long long idx = k > 0 ? k - 1 : n_pos + k;       // 1-based; negative = relative to the positions read so far
                    if (k == 0 || idx < 0 || idx >= n_pos) return fail("face vertex index out of bounds");
                    vs.push_back((uint32_t)idx);
                }
                for (size_t k = 1; k + 1 < vs.size(); k++) {                // fan triangulation
                    if (indices.size() + 3 > MAX_INDICES) return fail("too many faces");
                    indices.push_back(vs[0]); indices.push_back(vs[k]); indices.push_back(vs[k + 1]);
                }
            } else if (tl == 1 && (tb[0] == 'o' || tb[0] == 'g') && have_faces) {
                break;                                                      // models[0] only, src/mesh.rs:42
            }
        }
        p = eol < end ? eol + 1 : end;
    }
    if (!have_faces) return fail("no model in the obj file");               // `&models[0]` on an empty Vec panics, src/mesh.rs:42
    return true;
}

} // namespace rt
