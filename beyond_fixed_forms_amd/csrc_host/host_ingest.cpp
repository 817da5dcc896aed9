// libbff_host.so -- native host side of scene ingestion (no GPU code; loaded with ctypes.PyDLL).
//
// The reference keeps a scene's 2-D masks as a Python list of {"length": H*W, "counts": int ndarray of 1-based
// (start, len) pairs} (tools/utils/rle_encode_decode.py:10-32, written by tools/segmentation_2d.py:500) and its depth
// frames as one ndarray per frame.  Turning ~10^4 such dicts into flat run tables with NumPy costs ~50 ms per
// ScanNet200-size scene on one core; here the Python objects are visited once under the GIL (buffer pointers only)
// and the byte work -- run tables with the decode semantics of rle_decode_batch (RLE:45-57), frame copies into
// pinned staging -- runs on a few native threads with the GIL released.
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Piece { const void *ptr; int64_t n; int itemsize; int64_t length; Py_buffer view; bool held; };

template <typename F>
void parallel_for(int n_threads, int64_t n, F f)
{
    n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n));
    if (n_threads == 1) { f(0, n); return; }
    std::vector<std::thread> th;
    const int64_t per = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        const int64_t lo = t * per, hi = std::min(n, lo + per);
        if (lo < hi) th.emplace_back([=] { f(lo, hi); });
    }
    for (auto &x : th) x.join();
}

int64_t as_i64(const void *p, int itemsize, int64_t i)
{
    return itemsize == 8 ? reinterpret_cast<const int64_t *>(p)[i] : (int64_t)reinterpret_cast<const int32_t *>(p)[i];
}

}  // namespace

extern "C" {

// rles: sequence of dicts {"length", "counts"} (counts: C-contiguous int64 or int32 ndarray, even size).
// Writes, with the semantics of rle_decode_batch (counts cast to int32, `mask[start-1 : start-1+len] = 1`, python
// slicing clips at `length`):  run_start / run_end int32 [pairs] (0-based, end exclusive, clipped; empty runs stay in
// the table as start == end, which every consumer treats as nothing), mask_run_offs int32 [n + 1].
// Returns the number of runs, or  -1 bad object / dtype,  -2 capacity too small,  -3 odd number of counts,
// -4 a start < 1 (negative python slice in the reference),  -5 runs of one mask unsorted or overlapping (the caller
// normalises those on the slow path),  -6 a length different from expect_length (when expect_length > 0).
long long bff_host_pack_rles(PyObject *rles, int32_t *run_start, int32_t *run_end, long long cap, int32_t *offs,
                             long long expect_length, int n_threads)
{
    static PyObject *k_counts = PyUnicode_InternFromString("counts");     // interned once: PyDict_GetItem hashes no string
    static PyObject *k_length = PyUnicode_InternFromString("length");
    PyObject *seq = PySequence_Fast(rles, "rles must be a sequence");
    if (!seq) { PyErr_Clear(); return -1; }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    std::vector<Piece> pieces((size_t)n);
    long long total = 0, rc = 0;
    Py_ssize_t got = 0;
    for (; got < n; ++got) {
        PyObject *d = PySequence_Fast_GET_ITEM(seq, got);
        PyObject *counts = PyDict_Check(d) ? PyDict_GetItem(d, k_counts) : nullptr;      // borrowed
        PyObject *length = PyDict_Check(d) ? PyDict_GetItem(d, k_length) : nullptr;
        Piece &p = pieces[(size_t)got];
        p.held = false;
        if (!counts || !length || PyObject_GetBuffer(counts, &p.view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) {
            PyErr_Clear(); rc = -1; break;
        }
        p.held = true;
        const char *fmt = p.view.format ? p.view.format : "";
        const bool is_int = (fmt[0] == 'l' || fmt[0] == 'q' || fmt[0] == 'i') && fmt[1] == 0;
        if (!is_int || (p.view.itemsize != 8 && p.view.itemsize != 4)) { rc = -1; ++got; break; }
        p.ptr = p.view.buf; p.itemsize = (int)p.view.itemsize; p.n = p.view.len / p.view.itemsize;
        p.length = PyLong_AsLongLong(length);
        if (p.length == -1 && PyErr_Occurred()) { PyErr_Clear(); rc = -1; ++got; break; }
        if (p.n & 1) { rc = -3; ++got; break; }
        if (expect_length > 0 && p.length != expect_length) { rc = -6; ++got; break; }
        offs[got] = (int32_t)total;
        total += p.n / 2;
    }
    if (rc == 0 && total > cap) rc = -2;
    if (rc == 0) {
        offs[n] = (int32_t)total;
        std::atomic<int> err{0};
        Py_BEGIN_ALLOW_THREADS
        parallel_for(n_threads, n, [&](int64_t lo, int64_t hi) {
            for (int64_t g = lo; g < hi; ++g) {
                const Piece &p = pieces[(size_t)g];
                int32_t *rs = run_start + offs[g], *re = run_end + offs[g];
                int64_t prev_end = 0;
                for (int64_t k = 0; k < p.n / 2; ++k) {
                    const int64_t s = (int64_t)(int32_t)as_i64(p.ptr, p.itemsize, 2 * k) - 1;       // .astype(int32)
                    const int64_t l = (int64_t)(int32_t)as_i64(p.ptr, p.itemsize, 2 * k + 1);
                    if (s < 0) { err.store(4); return; }
                    int64_t e = std::min(s + l, p.length);
                    if (e <= s) {                                   // empty after clipping: keep the table monotone
                        rs[k] = re[k] = (int32_t)std::min(std::max(prev_end, (int64_t)0), p.length);
                        continue;
                    }
                    if (s < prev_end) { err.store(5); return; }
                    rs[k] = (int32_t)s; re[k] = (int32_t)e;
                    prev_end = e;
                }
            }
        });
        Py_END_ALLOW_THREADS
        if (err.load()) rc = -err.load();
    }
    for (Py_ssize_t i = 0; i < got; ++i)
        if (pieces[(size_t)i].held) PyBuffer_Release(&pieces[(size_t)i].view);
    Py_DECREF(seq);
    return rc < 0 ? rc : total;
}

// frames: sequence of C-contiguous buffers of exactly `bytes_each` bytes (e.g. the uint16 depth frames of a scene);
// copies frame i to dst + i * bytes_each on n_threads threads with the GIL released.  Returns the number of frames
// or -1 (bad object / size).
long long bff_host_pack_frames(PyObject *frames, void *dst, long long bytes_each, int n_threads)
{
    PyObject *seq = PySequence_Fast(frames, "frames must be a sequence");
    if (!seq) { PyErr_Clear(); return -1; }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    std::vector<Py_buffer> views((size_t)n);
    Py_ssize_t got = 0;
    long long rc = n;
    for (; got < n; ++got) {
        if (PyObject_GetBuffer(PySequence_Fast_GET_ITEM(seq, got), &views[(size_t)got], PyBUF_C_CONTIGUOUS) != 0) {
            PyErr_Clear(); rc = -1; break;
        }
        if (views[(size_t)got].len != bytes_each) { rc = -1; ++got; break; }
    }
    if (rc >= 0) {
        Py_BEGIN_ALLOW_THREADS
        parallel_for(n_threads, n, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i)
                std::memcpy(static_cast<char *>(dst) + i * bytes_each, views[(size_t)i].buf, (size_t)bytes_each);
        });
        Py_END_ALLOW_THREADS
    }
    for (Py_ssize_t i = 0; i < got; ++i) PyBuffer_Release(&views[(size_t)i]);
    Py_DECREF(seq);
    return rc;
}

// ---- 16-bit grayscale PNG -> uint16 frames (the depth frames of a ScanNet scene, P:431-433) -------------------------
// The reference decodes one PNG per frame with cv2.imread(IMREAD_UNCHANGED); 300-600 of them sit in front of every
// scene.  Here the whole batch is decoded on native threads with the GIL released, straight into the caller's (pinned)
// staging: chunk walk, one zlib inflate of the concatenated IDAT data, the five PNG row filters (None, Sub, Up, Average,
// Paeth; 2 bytes per pixel) and the big-endian -> host byte order swap.  Everything else -- another bit depth or colour
// type, interlacing, a size different from the batch's, a damaged file -- is DECLINED (status != 0) and left to the
// caller's general decoder; nothing is guessed.
namespace {

inline uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// status: 0 ok, 1 cannot read, 2 not a PNG / damaged chunk structure, 3 not 16-bit grayscale non-interlaced,
// 4 size differs from (h, w), 5 inflate failed / wrong amount of data, 6 bad filter byte
int decode_png_gray16(const char *path, int h, int w, uint16_t *out, std::vector<unsigned char> &file,
                      std::vector<unsigned char> &idat, std::vector<unsigned char> &raw)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return 1;
    std::fseek(f, 0, SEEK_END);
    const long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (size < 8 + 25 + 12) { std::fclose(f); return 2; }
    file.resize((size_t)size);
    const size_t got = std::fread(file.data(), 1, (size_t)size, f);
    std::fclose(f);
    if (got != (size_t)size) return 1;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (std::memcmp(file.data(), sig, 8) != 0) return 2;
    size_t at = 8;
    bool have_ihdr = false, ended = false;
    idat.clear();
    while (at + 12 <= file.size() && !ended) {
        const uint32_t len = be32(&file[at]);
        const unsigned char *type = &file[at + 4];
        if (at + 12 + (size_t)len > file.size()) return 2;
        const unsigned char *data = &file[at + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return 2;
            const uint32_t pw = be32(data), ph = be32(data + 4);
            if (data[8] != 16 || data[9] != 0 || data[10] != 0 || data[11] != 0 || data[12] != 0) return 3;
            if ((int64_t)pw != w || (int64_t)ph != h) return 4;
            have_ihdr = true;
        } else if (!std::memcmp(type, "IDAT", 4)) {
            if (!have_ihdr) return 2;
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            ended = true;
        }
        at += 12 + (size_t)len;                       // length, type, data, crc (the crc is not verified, as cv2 does not either)
    }
    if (!have_ihdr || !ended || idat.empty()) return 2;
    const size_t stride = (size_t)w * 2, need = (stride + 1) * (size_t)h;
    raw.resize(need);
    uLongf out_len = (uLongf)need;
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != need) return 5;
    // unfilter (bpp = 2) and swap into the output in one pass per row.  Sub / Average / Paeth carry a dependency from
    // pixel to pixel; the two bytes of a pixel are independent chains, kept in registers (left and upper-left pixel)
    std::vector<unsigned char> &zero = file;              // the file's bytes are no longer needed: a zero row for y = 0
    zero.assign(stride, 0);
    const unsigned char *prev = zero.data();
    for (int y = 0; y < h; ++y) {
        unsigned char *row = raw.data() + (size_t)y * (stride + 1);
        const int ft = row[0];
        unsigned char *cur = row + 1;
        uint16_t *dst = out + (size_t)y * (size_t)w;
        unsigned a0 = 0, a1 = 0, c0 = 0, c1 = 0;
        switch (ft) {
        case 0:
            for (int x = 0; x < w; ++x) dst[x] = (uint16_t)(((unsigned)cur[2 * x] << 8) | cur[2 * x + 1]);
            break;
        case 1:
            for (int x = 0; x < w; ++x) {
                a0 = (a0 + cur[2 * x]) & 255u; a1 = (a1 + cur[2 * x + 1]) & 255u;
                cur[2 * x] = (unsigned char)a0; cur[2 * x + 1] = (unsigned char)a1;
                dst[x] = (uint16_t)((a0 << 8) | a1);
            }
            break;
        case 2:
            for (size_t i = 0; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + prev[i]);
            for (int x = 0; x < w; ++x) dst[x] = (uint16_t)(((unsigned)cur[2 * x] << 8) | cur[2 * x + 1]);
            break;
        case 3:
            for (int x = 0; x < w; ++x) {
                a0 = (cur[2 * x] + ((a0 + prev[2 * x]) >> 1)) & 255u;
                a1 = (cur[2 * x + 1] + ((a1 + prev[2 * x + 1]) >> 1)) & 255u;
                cur[2 * x] = (unsigned char)a0; cur[2 * x + 1] = (unsigned char)a1;
                dst[x] = (uint16_t)((a0 << 8) | a1);
            }
            break;
        case 4: {
            auto paeth = [](int a, int b, int c) {
                const int p = b - c, q = a - c;                                  // pa = |p|, pb = |q|, pc = |p + q|
                const int pa = p < 0 ? -p : p, pb = q < 0 ? -q : q, pc = (p + q) < 0 ? -(p + q) : (p + q);
                return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            };
            for (int x = 0; x < w; ++x) {
                const int b0 = prev[2 * x], b1 = prev[2 * x + 1];
                a0 = (unsigned)(cur[2 * x] + paeth((int)a0, b0, (int)c0)) & 255u;
                a1 = (unsigned)(cur[2 * x + 1] + paeth((int)a1, b1, (int)c1)) & 255u;
                cur[2 * x] = (unsigned char)a0; cur[2 * x + 1] = (unsigned char)a1;
                c0 = (unsigned)b0; c1 = (unsigned)b1;
                dst[x] = (uint16_t)((a0 << 8) | a1);
            }
            break;
        }
        default: return 6;
        }
        prev = cur;
    }
    return 0;
}

}  // namespace

// (height, width) of a PNG's IHDR into hw[2]; returns 0, or the decline status of decode_png_gray16 (3: the file is a
// PNG of another kind).
int bff_host_png_size(const char *path, int32_t *hw)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return 1;
    unsigned char head[33];
    const size_t got = std::fread(head, 1, sizeof(head), f);
    std::fclose(f);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (got != sizeof(head) || std::memcmp(head, sig, 8) != 0 || std::memcmp(head + 12, "IHDR", 4) != 0) return 2;
    hw[1] = (int32_t)be32(head + 16);
    hw[0] = (int32_t)be32(head + 20);
    if (head[24] != 16 || head[25] != 0 || head[26] != 0 || head[27] != 0 || head[28] != 0) return 3;
    return 0;
}

// paths: sequence of str (or bytes) file names; frame i is decoded to dst + i * h * w (uint16, host byte order); status[i]
// = 0 or the reason it was declined (its frame is then untouched).  Returns the number of frames decoded, -1 on a bad
// argument.
long long bff_host_decode_depth_pngs(PyObject *paths, uint16_t *dst, int h, int w, int32_t *status, int n_threads)
{
    PyObject *seq = PySequence_Fast(paths, "paths must be a sequence");
    if (!seq) { PyErr_Clear(); return -1; }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    std::vector<std::string> names((size_t)n);
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *o = PySequence_Fast_GET_ITEM(seq, i);
        PyObject *b = PyUnicode_Check(o) ? PyUnicode_EncodeFSDefault(o) : (PyBytes_Check(o) ? (Py_INCREF(o), o) : nullptr);
        if (!b) { PyErr_Clear(); Py_DECREF(seq); return -1; }
        names[(size_t)i] = PyBytes_AS_STRING(b);
        Py_DECREF(b);
    }
    Py_DECREF(seq);
    if (h <= 0 || w <= 0 || !dst || !status) return -1;
    std::atomic<long long> ok{0};
    Py_BEGIN_ALLOW_THREADS
    parallel_for(n_threads, n, [&](int64_t lo, int64_t hi) {
        std::vector<unsigned char> file, idat, raw;                  // per thread, reused frame after frame
        for (int64_t i = lo; i < hi; ++i) {
            status[i] = decode_png_gray16(names[(size_t)i].c_str(), h, w, dst + (size_t)i * (size_t)h * (size_t)w, file, idat, raw);
            if (status[i] == 0) ok.fetch_add(1);
        }
    });
    Py_END_ALLOW_THREADS
    return ok.load();
}

// Concatenates n host buffers (ptrs[i], bytes[i]) into dst: the per-frame confidence tensors of a mask_2d file.  One
// call instead of one torch / numpy call per frame -- every ATen call from Python hands the GIL over and back, which
// with four loader threads and the compute thread contending cost 30-60 us per call (300 frames: 8-18 ms per scene).
long long bff_host_gather_bytes(const long long *ptrs, const long long *bytes, long long n, void *dst)
{
    long long at = 0;
    for (long long i = 0; i < n; ++i) {
        if (bytes[i] < 0 || (bytes[i] > 0 && !ptrs[i])) return -1;
        std::memcpy(static_cast<char *>(dst) + at, reinterpret_cast<const void *>(ptrs[i]), (size_t)bytes[i]);
        at += bytes[i];
    }
    return at;
}

int bff_host_abi(void) { return 3; }

}  // extern "C"
