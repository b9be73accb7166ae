/* Host-side half of the learner's ingest when experiences arrive as Python objects from CPU agents
 * (surreal/learner/aggregator.py:106-262: MultistepAggregatorWithInfo.aggregate walks B x N per-step arrays with nested
 * Python loops, 1.8 s for the 1024 x 128 x 376 batch; numpy's stack over the 131 072 leaves: 0.16 s).
 *
 * fill(out, exp_list, field, path, per_step) writes ONE field of the batch straight into `out` -- normally a pinned
 * struct-of-arrays staging buffer (PinnedBatchStager.host_views) -- in two phases:
 *   1. with the GIL: walk exp_list -> exp[field] (-> every step) -> path (dict keys / sequence indices) and note where
 *      each leaf's bytes are (ndarray data pointers; Python scalars are converted here);
 *   2. without the GIL: copy / convert the leaves into their slots, split over a few OpenMP threads.
 * Leaves: C-contiguous float32 / float64 / uint8 / bool ndarrays whose size is the slot's, or Python float / int / bool
 * for one-element slots.  Anything else returns -1 and the caller takes the numpy path for that field (nothing has been
 * written to `out` by then: phase 1 validates everything first).
 * CPython + numpy C API; built by surreal_amd.build with gcc -fopenmp.  No device code, no torch. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    const void* src;      /* leaf bytes, or NULL: `scalar` holds the value */
    double scalar;
    int kind;             /* 0 same dtype (memcpy), 1 float64 -> float32, 2 bool / uint8 -> float32 */
} leaf_t;

static PyObject* follow(PyObject* o, PyObject* path) {
    /* borrowed reference to o[path[0]][path[1]]...; NULL (no exception set) when a step is missing */
    const Py_ssize_t n = PyTuple_GET_SIZE(path);
    for (Py_ssize_t i = 0; i < n && o; ++i) {
        PyObject* k = PyTuple_GET_ITEM(path, i);
        if (PyDict_Check(o)) {
            o = PyDict_GetItemWithError(o, k);
        } else if (PyList_Check(o) && PyLong_Check(k)) {
            const Py_ssize_t j = PyLong_AsSsize_t(k);
            o = (j >= 0 && j < PyList_GET_SIZE(o)) ? PyList_GET_ITEM(o, j) : NULL;
        } else if (PyTuple_Check(o) && PyLong_Check(k)) {
            const Py_ssize_t j = PyLong_AsSsize_t(k);
            o = (j >= 0 && j < PyTuple_GET_SIZE(o)) ? PyTuple_GET_ITEM(o, j) : NULL;
        } else {
            o = NULL;
        }
    }
    return o;
}

/* -> 0 ok, -1 unsupported leaf */
static int note_leaf(PyObject* v, int out_type, npy_intp slot_elems, leaf_t* L) {
    L->src = NULL; L->scalar = 0.0; L->kind = 0;
    if (PyArray_Check(v)) {
        PyArrayObject* a = (PyArrayObject*)v;
        if (!PyArray_IS_C_CONTIGUOUS(a) || PyArray_SIZE(a) != slot_elems) return -1;
        const int t = PyArray_TYPE(a);
        L->src = PyArray_DATA(a);
        if (t == out_type) { L->kind = 0; return 0; }
        if (out_type == NPY_FLOAT32 && t == NPY_FLOAT64) { L->kind = 1; return 0; }
        if (out_type == NPY_FLOAT32 && (t == NPY_BOOL || t == NPY_UINT8)) { L->kind = 2; return 0; }
        return -1;
    }
    if (slot_elems != 1) return -1;
    if (PyFloat_Check(v)) { L->scalar = PyFloat_AS_DOUBLE(v); return 0; }
    if (PyBool_Check(v)) { L->scalar = (v == Py_True) ? 1.0 : 0.0; return 0; }
    if (PyLong_Check(v)) { L->scalar = (double)PyLong_AsLong(v); return PyErr_Occurred() ? -1 : 0; }
    if (PyArray_IsScalar(v, Generic)) {
        PyObject* f = PyNumber_Float(v);
        if (!f) { PyErr_Clear(); return -1; }
        L->scalar = PyFloat_AS_DOUBLE(f);
        Py_DECREF(f);
        return 0;
    }
    return -1;
}

static PyObject* smx_fill(PyObject* self, PyObject* args) {
    PyObject *out_o, *exps, *field, *path;
    int per_step = 1, threads = 4;
    if (!PyArg_ParseTuple(args, "OOOO!|ii", &out_o, &exps, &field, &PyTuple_Type, &path, &per_step, &threads)) return NULL;
    if (!PyArray_Check(out_o) || !PyList_Check(exps)) {
        PyErr_SetString(PyExc_TypeError, "fill(out: ndarray, exp_list: list, field, path: tuple[, per_step, threads])");
        return NULL;
    }
    PyArrayObject* out = (PyArrayObject*)out_o;
    const int out_type = PyArray_TYPE(out);
    if (!PyArray_IS_C_CONTIGUOUS(out) || !PyArray_ISWRITEABLE(out) || (out_type != NPY_FLOAT32 && out_type != NPY_UINT8))
        return PyLong_FromLong(-1);
    const Py_ssize_t B = PyList_GET_SIZE(exps);
    const int nd = PyArray_NDIM(out);
    if (nd < 1 || PyArray_DIM(out, 0) != B || (per_step && nd < 2)) return PyLong_FromLong(-1);
    const npy_intp N = per_step ? PyArray_DIM(out, 1) : 1;
    npy_intp slot = 1;
    for (int d = per_step ? 2 : 1; d < nd; ++d) slot *= PyArray_DIM(out, d);
    const npy_intp total = (npy_intp)B * N;
    leaf_t* L = (leaf_t*)PyMem_Malloc(sizeof(leaf_t) * (size_t)(total > 0 ? total : 1));
    if (!L) return PyErr_NoMemory();
    /* ---- phase 1 (GIL held): locate every leaf ---- */
    int bad = 0;
    for (Py_ssize_t b = 0; b < B && !bad; ++b) {
        PyObject* e = PyList_GET_ITEM(exps, b);
        PyObject* v = PyDict_Check(e) ? PyDict_GetItemWithError(e, field) : NULL;
        if (!v) { bad = 1; break; }
        if (!per_step) {
            PyObject* leaf = follow(v, path);
            if (!leaf || note_leaf(leaf, out_type, slot, &L[b])) bad = 1;
            continue;
        }
        if (PyArray_Check(v) && PyTuple_GET_SIZE(path) == 0) {
            /* the whole (N, ...) field as one array (what a device-tier replay hands over) */
            PyArrayObject* a = (PyArrayObject*)v;
            if (!PyArray_IS_C_CONTIGUOUS(a) || PyArray_SIZE(a) != N * slot) { bad = 1; break; }
            const int t = PyArray_TYPE(a);
            int kind;
            if (t == out_type) kind = 0;
            else if (out_type == NPY_FLOAT32 && t == NPY_FLOAT64) kind = 1;
            else if (out_type == NPY_FLOAT32 && (t == NPY_BOOL || t == NPY_UINT8)) kind = 2;
            else { bad = 1; break; }
            const size_t es = (size_t)PyArray_ITEMSIZE(a);
            for (npy_intp s = 0; s < N; ++s) {
                L[b * N + s].src = (const char*)PyArray_DATA(a) + (size_t)s * slot * es;
                L[b * N + s].kind = kind;
                L[b * N + s].scalar = 0.0;
            }
            continue;
        }
        const int is_list = PyList_Check(v), is_tuple = PyTuple_Check(v);
        if (!(is_list || is_tuple) || (is_list ? PyList_GET_SIZE(v) : PyTuple_GET_SIZE(v)) != N) { bad = 1; break; }
        for (npy_intp s = 0; s < N; ++s) {
            PyObject* step = is_list ? PyList_GET_ITEM(v, s) : PyTuple_GET_ITEM(v, s);
            PyObject* leaf = follow(step, path);
            if (!leaf || note_leaf(leaf, out_type, slot, &L[b * N + s])) { bad = 1; break; }
        }
    }
    if (PyErr_Occurred()) { PyMem_Free(L); return NULL; }
    if (bad) { PyMem_Free(L); return PyLong_FromLong(-1); }
    /* ---- phase 2 (GIL released): copy / convert ---- */
    char* dst = (char*)PyArray_DATA(out);
    const size_t osz = out_type == NPY_FLOAT32 ? 4 : 1;
    if (threads < 1) threads = 1;
    if (threads > 16) threads = 16;
    if ((size_t)total * (size_t)slot * osz < (1u << 20)) threads = 1;
    Py_BEGIN_ALLOW_THREADS
#pragma omp parallel for schedule(static) num_threads(threads)
    for (npy_intp i = 0; i < total; ++i) {
        char* d = dst + (size_t)i * (size_t)slot * osz;
        const leaf_t* l = &L[i];
        if (!l->src) {
            if (out_type == NPY_FLOAT32) *(float*)d = (float)l->scalar; else *(uint8_t*)d = (uint8_t)l->scalar;
        } else if (l->kind == 0) {
            memcpy(d, l->src, (size_t)slot * osz);
        } else if (l->kind == 1) {
            const double* s = (const double*)l->src;
            float* f = (float*)d;
            for (npy_intp k = 0; k < slot; ++k) f[k] = (float)s[k];
        } else {
            const uint8_t* s = (const uint8_t*)l->src;
            float* f = (float*)d;
            for (npy_intp k = 0; k < slot; ++k) f[k] = (float)s[k];
        }
    }
    Py_END_ALLOW_THREADS
    PyMem_Free(L);
    return PyLong_FromSsize_t((Py_ssize_t)total);
}

static PyMethodDef methods[] = {
    {"fill", smx_fill, METH_VARARGS,
     "fill(out, exp_list, field, path, per_step=1, threads=4) -> leaves written, or -1 (unsupported layout: nothing written)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_smx_host", "host-side batch assembly (see smx_host.c)", -1, methods};

PyMODINIT_FUNC PyInit__smx_host(void) {
    import_array();
    return PyModule_Create(&moddef);
}
