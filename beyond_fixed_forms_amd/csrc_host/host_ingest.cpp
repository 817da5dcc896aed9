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

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
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

int bff_host_abi(void) { return 1; }

}  // extern "C"
