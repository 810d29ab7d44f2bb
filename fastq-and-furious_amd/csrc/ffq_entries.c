/* ffq_entries.c -- the default entryfunc of readfastq_iter, applied to a whole offset table.
 *
 * The reference builds one entry per scanner call,
 *     entryfunc(buf, pos, globaloffset) = (buf[pos[0]+1:pos[1]], buf[pos[2]:pos[3]], buf[pos[4]:pos[5]])
 * (/root/reference/src/fastqandfurious.py:161-171), in the interpreter.  Here the scanner hands back
 * the table of a whole buffer fill; this module cuts the three slices of every row in one call and
 * returns the list of tuples the iterator then yields from.  Host glue only (CPython C API, as the
 * reference's own extension is): nothing of the scan runs here, and readfastq_iter falls back to
 * the same slices in Python when the module is not built.
 *
 *     entries(buf, rows, shift=0, hskip=1, cls=None) -> [(header, sequence, quality), ...]
 *
 * buf    any C-contiguous buffer of bytes (bytes, memoryview, the pinned fill of the stream front end)
 * rows   C-contiguous buffer of int64, six per record (the table of ffq_scan_*)
 * shift  subtracted from every position first (rows in stream coordinates, buf one fill of it)
 * hskip  the header slice starts at pos[0] + hskip: 1 drops the '@' as entryfunc does; 0 keeps it, as
 *        the reference's index replay does (/root/reference/src/demo/benchmark.py:62-71)
 * cls    a tuple subclass with three fields and no state of its own (the reference's `Entry`
 *        namedtuple, entryfunc_namedtuple, :146-158): the entries are instances of it
 *
 * Slices follow Python's rules (negative positions count from the end, bounds are clamped, an
 * inverted range is empty), so the result equals the Python expression above for ANY row.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

/* A tuple of three bytes / array('b') objects refers to nothing the cycle collector looks after: it cannot be part of a
 * cycle, and the collector itself takes such tuples off its lists when it first meets them.  It meets these while the
 * list is still being built (a collection every 700 allocations): hundreds of thousands of live young tuples are then
 * promoted from generation to generation, which makes FULL collections due (the 25 % rule counts promoted objects) -- a
 * dozen per 800 000 records, milliseconds each in a process with many objects (torch imported: 188 ms per 800 000 records
 * with the collector on, 110 with it off).  Untracked at birth they are never promoted. */
#define UNTRACK(t) PyObject_GC_UnTrack(t)

static PyObject *cut(const char *base, Py_ssize_t len, int64_t a, int64_t b)
{
    Py_ssize_t start = (Py_ssize_t)a, stop = (Py_ssize_t)b;
    const Py_ssize_t n = PySlice_AdjustIndices(len, &start, &stop, 1);
    return PyBytes_FromStringAndSize(base + start, n);
}

static PyObject *entries(PyObject *self, PyObject *args)
{
    Py_buffer buf, rows;
    long long shift = 0, hskip = 1;
    PyObject *cls = Py_None;
    (void)self;
    if (!PyArg_ParseTuple(args, "y*y*|LLO", &buf, &rows, &shift, &hskip, &cls)) return NULL;
    PyObject *list = NULL;
    PyTypeObject *tp = &PyTuple_Type;
    if (cls != Py_None) {
        if (!PyType_Check(cls) || !PyType_IsSubtype((PyTypeObject *)cls, &PyTuple_Type) ||
            ((PyTypeObject *)cls)->tp_basicsize != PyTuple_Type.tp_basicsize || ((PyTypeObject *)cls)->tp_itemsize != PyTuple_Type.tp_itemsize) {
            PyErr_SetString(PyExc_TypeError, "cls must be a tuple subclass without fields of its own (a namedtuple)");
            goto done;
        }
        tp = (PyTypeObject *)cls;
    }
    if (rows.len % 48 != 0 || (rows.itemsize != 8 && rows.itemsize != 1)) {
        PyErr_SetString(PyExc_ValueError, "rows must hold six int64 positions per record");
        goto done;
    }
    {
        const Py_ssize_t n = rows.len / 48;
        const int64_t *p = (const int64_t *)rows.buf;
        const char *base = (const char *)buf.buf;
        list = PyList_New(n);
        if (!list) goto done;
        for (Py_ssize_t i = 0; i < n; i++, p += 6) {
            PyObject *h = cut(base, buf.len, p[0] - shift + hskip, p[1] - shift);
            PyObject *s = cut(base, buf.len, p[2] - shift, p[3] - shift);
            PyObject *q = cut(base, buf.len, p[4] - shift, p[5] - shift);
            PyObject *t = !(h && s && q) ? NULL : (tp == &PyTuple_Type) ? PyTuple_New(3) : tp->tp_alloc(tp, 3);
            if (!t) {
                Py_XDECREF(h); Py_XDECREF(s); Py_XDECREF(q);
                Py_CLEAR(list);
                goto done;
            }
            PyTuple_SET_ITEM(t, 0, h);
            PyTuple_SET_ITEM(t, 1, s);
            PyTuple_SET_ITEM(t, 2, q);
            UNTRACK(t);
            PyList_SET_ITEM(list, i, t);
        }
    }
done:
    PyBuffer_Release(&buf);
    PyBuffer_Release(&rows);
    return list;
}

/* entries_phred(buf, rows, shift, qual, qoff, array_type) -> [(header, sequence, array('b')), ...]
 * What the reference's documented decode builds per record in an entryfunc of the user's own,
 *     quality = array('b'); quality.frombytes(buf[pos[4]:pos[5]]); arrayadd_b(quality, -33)
 * (/root/reference/doc/user-guide.rst:126-141, :206-214), for a whole table at once: the decoded
 * bytes come from the stream's bulk decode on the device -- qual (int8) with the offsets qoff (int64,
 * one more than rows: where each record's bytes start, and where the last one's end) -- and are only
 * wrapped here.  array_type: array.array.                                                         */
/* How a record's array('b') is made.  The array module has no C API; what Python offers is the constructor (a call with a
 * typecode string per record) or slicing ONE big array (a slice object + the subscript protocol per record: what round 5
 * did, out of a process-global array that a second thread's call could overwrite mid-loop -- the round-5 advisor's finding).
 * Round 6: the records' arrays are built DIRECTLY -- tp_alloc of the array type, the items from PyMem_Malloc, the
 * descriptor taken from an array('b') the real constructor made -- which is what arraymodule.c's newarrayobject does, with
 * no global state at all.  That relies on CPython's private `arrayobject` layout, so it is used only when a probe at first
 * use finds every field where this file expects it (size, item pointer, allocation, export count observed through the
 * public buffer protocol); otherwise -- another CPython -- the records are slices of a PER-CALL array.               */
typedef struct {
    PyObject_VAR_HEAD
    char *ob_item;
    Py_ssize_t allocated;
    const void *ob_descr;
    PyObject *weakreflist;
    Py_ssize_t ob_exports;
} ffq_arrayobject;

static int g_direct = -1;                 /* -1: not probed yet; 0: slices of a per-call array; 1: direct construction */
static const void *g_descr_b = NULL;      /* array('b')'s type descriptor (static data of the array module) */
static PyTypeObject *g_direct_type = NULL;

/* array_type('b', bytes-of-n) through the public constructor */
static PyObject *array_from(PyObject *atype, const char *src, Py_ssize_t n)
{
    PyObject *raw = PyBytes_FromStringAndSize(src, n);
    PyObject *a = raw ? PyObject_CallFunction(atype, "sO", "b", raw) : NULL;
    Py_XDECREF(raw);
    return a;
}

static void probe_direct(PyObject *atype)
{
    g_direct = 0;
#if PY_VERSION_HEX >= 0x03080000 && PY_VERSION_HEX < 0x030D0000 && !defined(PYPY_VERSION)
    if (!PyType_Check(atype)) return;
    PyTypeObject *tp = (PyTypeObject *)atype;
    if (tp->tp_basicsize != (Py_ssize_t)sizeof(ffq_arrayobject) || tp->tp_itemsize != 0) return;
    static const char pat[7] = {1, -2, 3, -4, 5, -6, 7};
    PyObject *a = array_from(atype, pat, 7);
    if (!a) { PyErr_Clear(); return; }
    ffq_arrayobject *ao = (ffq_arrayobject *)a;
    Py_buffer v;
    int ok = Py_TYPE(a) == tp && Py_SIZE(a) == 7 && ao->allocated >= 7 && ao->ob_item && ao->ob_exports == 0 && ao->weakreflist == NULL &&
             memcmp(ao->ob_item, pat, 7) == 0 && ao->ob_descr != NULL;
    if (ok && PyObject_GetBuffer(a, &v, PyBUF_SIMPLE) == 0) {
        ok = v.buf == (void *)ao->ob_item && v.len == 7 && v.itemsize == 1 && ao->ob_exports == 1;
        PyBuffer_Release(&v);
        ok = ok && ao->ob_exports == 0;
    } else { PyErr_Clear(); ok = 0; }
    if (ok) {
        /* one built by hand must BE an array: equal to the constructor's, typecode 'b', growable, freed by the type's own dealloc */
        ffq_arrayobject *h = (ffq_arrayobject *)tp->tp_alloc(tp, 0);
        if (h) {
            h->ob_item = (char *)PyMem_Malloc(7);
            if (h->ob_item) {
                memcpy(h->ob_item, pat, 7);
                h->allocated = 7; h->ob_descr = ao->ob_descr; h->weakreflist = NULL; h->ob_exports = 0;
                Py_SET_SIZE(h, 7);
                PyObject *tc = PyObject_GetAttrString((PyObject *)h, "typecode");
                const int same = PyObject_RichCompareBool((PyObject *)h, a, Py_EQ);
                PyObject *r = PyObject_CallMethod((PyObject *)h, "append", "i", -8);
                ok = same == 1 && tc && PyUnicode_Check(tc) && PyUnicode_CompareWithASCIIString(tc, "b") == 0 && r && Py_SIZE(h) == 8 &&
                     h->ob_item[7] == -8;
                Py_XDECREF(tc); Py_XDECREF(r);
                if (PyErr_Occurred()) { PyErr_Clear(); ok = 0; }
            } else ok = 0;
            Py_DECREF((PyObject *)h);
        } else { PyErr_Clear(); ok = 0; }
    }
    if (ok) { g_direct = 1; g_descr_b = ao->ob_descr; g_direct_type = tp; Py_INCREF(atype); }
    Py_DECREF(a);
#else
    (void)atype;
#endif
}

static inline PyObject *array_direct(const char *src, Py_ssize_t n)
{
    ffq_arrayobject *h = (ffq_arrayobject *)g_direct_type->tp_alloc(g_direct_type, 0);
    if (!h) return NULL;
    h->ob_item = NULL;
    if (n > 0) {
        h->ob_item = (char *)PyMem_Malloc((size_t)n);
        if (!h->ob_item) { Py_DECREF((PyObject *)h); return PyErr_NoMemory(); }
        memcpy(h->ob_item, src, (size_t)n);
    }
    h->allocated = n; h->ob_descr = g_descr_b; h->weakreflist = NULL; h->ob_exports = 0;
    Py_SET_SIZE(h, n);
    /* (CPython 3.10's array is a heap type and as such a GC type: its traverse visits the type alone -- nothing a cycle
     * could go through -- so the records' arrays need not sit on the collector's lists, like their tuples) */
    if (PyType_IS_GC(g_direct_type)) PyObject_GC_UnTrack((PyObject *)h);
    return (PyObject *)h;
}

static PyObject *entries_phred(PyObject *self, PyObject *args)
{
    Py_buffer buf, rows, qual, qoff;
    long long shift = 0;
    PyObject *atype = NULL, *big = NULL;
    (void)self;
    if (!PyArg_ParseTuple(args, "y*y*Ly*y*O", &buf, &rows, &shift, &qual, &qoff, &atype)) return NULL;
    PyObject *list = NULL;
    const Py_ssize_t n = rows.len / 48;
    if (rows.len % 48 != 0 || qoff.len < (n + 1) * 8) {
        PyErr_SetString(PyExc_ValueError, "rows must hold six int64 positions per record and qoff one offset more than rows");
        goto done;
    }
    if (n > 0) {
        const int64_t *p = (const int64_t *)rows.buf;
        const int64_t *o = (const int64_t *)qoff.buf;
        const char *base = (const char *)buf.buf;
        const int64_t q0 = o[0];
        if (q0 < 0 || o[n] < q0 || o[n] > (int64_t)qual.len) {
            PyErr_SetString(PyExc_ValueError, "quality offsets do not fit the decoded stream");
            goto done;
        }
        if (g_direct < 0 || (g_direct == 1 && (PyObject *)g_direct_type != atype)) probe_direct(atype);
        const int direct = g_direct == 1 && (PyObject *)g_direct_type == atype;
        if (!direct) {
            /* the decoded bytes of all these rows as ONE array('b') of this call's own; a record's array is a slice of it */
            big = array_from(atype, (const char *)qual.buf + q0, (Py_ssize_t)(o[n] - q0));
            if (!big) goto done;
        }
        list = PyList_New(n);
        if (!list) goto done;
        for (Py_ssize_t i = 0; i < n; i++, p += 6) {
            /* record i's bytes start at o[i] and are pos5 - pos4 of them (include/ffq.h: the packed stream and the
             * segmented layout of FFQ_F_SINGLE_PASS alike; o[i + 1] - o[i] is a length only in the packed one) */
            const int64_t ln = p[5] - p[4];
            if (o[i] < q0 || ln < 0 || o[i] + ln > o[n]) {
                PyErr_SetString(PyExc_ValueError, "quality offsets do not fit the decoded stream");
                Py_CLEAR(list);
                goto done;
            }
            PyObject *h = cut(base, buf.len, p[0] - shift + 1, p[1] - shift);
            PyObject *s = cut(base, buf.len, p[2] - shift, p[3] - shift);
            PyObject *q = direct ? array_direct((const char *)qual.buf + o[i], (Py_ssize_t)ln)
                                 : PySequence_GetSlice(big, (Py_ssize_t)(o[i] - q0), (Py_ssize_t)(o[i] + ln - q0));
            PyObject *t = !(h && s && q) ? NULL : PyTuple_New(3);
            if (!t) {
                Py_XDECREF(h); Py_XDECREF(s); Py_XDECREF(q);
                Py_CLEAR(list);
                goto done;
            }
            PyTuple_SET_ITEM(t, 0, h);
            PyTuple_SET_ITEM(t, 1, s);
            PyTuple_SET_ITEM(t, 2, q);
            UNTRACK(t);
            PyList_SET_ITEM(list, i, t);
        }
    }
    else list = PyList_New(0);
done:
    Py_XDECREF(big);
    PyBuffer_Release(&buf);
    PyBuffer_Release(&rows);
    PyBuffer_Release(&qual);
    PyBuffer_Release(&qoff);
    return list;
}

/* sparse(n_total, index, col, coloff) -> list of n_total items: bytes(col[coloff[i] : coloff[i + 1]]) at position
 * index[i], None everywhere else.  What an entryfunc that filters -- the reference's user guide,
 * /root/reference/doc/user-guide.rst:153-180: `buf[posarray[2]:posarray[3]] if posarray[3] - posarray[2] < THRESHOLD else
 * None` -- makes readfastq_iter yield for a whole buffer fill, from the stream's device-side selection (kept rows' ordinals)
 * and gathered column: a dropped record costs one pointer in a list.
 * sparse_entries(n_total, index, buf, rows, shift, cls=None): the same with the (header, sequence, quality) tuple of every
 * kept row (rows: the KEPT rows, six int64 each).                                                                      */
static PyObject *sparse(PyObject *self, PyObject *args)
{
    Py_buffer idx, col, off;
    Py_ssize_t n_total = 0;
    (void)self;
    if (!PyArg_ParseTuple(args, "ny*y*y*", &n_total, &idx, &col, &off)) return NULL;
    PyObject *list = NULL;
    const Py_ssize_t k = idx.len / 8;
    if (n_total < 0 || idx.len % 8 != 0 || off.len < (k + 1) * 8) {
        PyErr_SetString(PyExc_ValueError, "index must hold int64 ordinals and coloff one offset more");
        goto done;
    }
    list = PyList_New(n_total);
    if (!list) goto done;
    for (Py_ssize_t i = 0; i < n_total; i++) { Py_INCREF(Py_None); PyList_SET_ITEM(list, i, Py_None); }
    {
        const int64_t *ix = (const int64_t *)idx.buf, *o = (const int64_t *)off.buf;
        for (Py_ssize_t i = 0; i < k; i++) {
            if (ix[i] < 0 || ix[i] >= n_total || o[i] < 0 || o[i + 1] < o[i] || o[i + 1] > (int64_t)col.len) {
                PyErr_SetString(PyExc_ValueError, "selection does not fit the fill");
                Py_CLEAR(list);
                goto done;
            }
            PyObject *b = PyBytes_FromStringAndSize((const char *)col.buf + o[i], (Py_ssize_t)(o[i + 1] - o[i]));
            if (!b) { Py_CLEAR(list); goto done; }
            PyObject *old = PyList_GET_ITEM(list, (Py_ssize_t)ix[i]);
            PyList_SET_ITEM(list, (Py_ssize_t)ix[i], b);
            Py_DECREF(old);
        }
    }
done:
    PyBuffer_Release(&idx); PyBuffer_Release(&col); PyBuffer_Release(&off);
    return list;
}

static PyObject *sparse_entries(PyObject *self, PyObject *args)
{
    Py_buffer idx, buf, rows;
    Py_ssize_t n_total = 0;
    long long shift = 0;
    PyObject *cls = Py_None;
    (void)self;
    if (!PyArg_ParseTuple(args, "ny*y*y*L|O", &n_total, &idx, &buf, &rows, &shift, &cls)) return NULL;
    PyObject *list = NULL;
    PyTypeObject *tp = &PyTuple_Type;
    const Py_ssize_t k = idx.len / 8;
    if (cls != Py_None) {
        if (!PyType_Check(cls) || !PyType_IsSubtype((PyTypeObject *)cls, &PyTuple_Type) ||
            ((PyTypeObject *)cls)->tp_basicsize != PyTuple_Type.tp_basicsize || ((PyTypeObject *)cls)->tp_itemsize != PyTuple_Type.tp_itemsize) {
            PyErr_SetString(PyExc_TypeError, "cls must be a tuple subclass without fields of its own (a namedtuple)");
            goto done;
        }
        tp = (PyTypeObject *)cls;
    }
    if (n_total < 0 || idx.len % 8 != 0 || rows.len != k * 48) {
        PyErr_SetString(PyExc_ValueError, "index must hold one int64 ordinal per kept row, rows six int64 positions");
        goto done;
    }
    list = PyList_New(n_total);
    if (!list) goto done;
    for (Py_ssize_t i = 0; i < n_total; i++) { Py_INCREF(Py_None); PyList_SET_ITEM(list, i, Py_None); }
    {
        const int64_t *ix = (const int64_t *)idx.buf, *p = (const int64_t *)rows.buf;
        const char *base = (const char *)buf.buf;
        for (Py_ssize_t i = 0; i < k; i++, p += 6) {
            if (ix[i] < 0 || ix[i] >= n_total) {
                PyErr_SetString(PyExc_ValueError, "selection does not fit the fill");
                Py_CLEAR(list);
                goto done;
            }
            PyObject *h = cut(base, buf.len, p[0] - shift + 1, p[1] - shift);
            PyObject *q2 = cut(base, buf.len, p[2] - shift, p[3] - shift);
            PyObject *q = cut(base, buf.len, p[4] - shift, p[5] - shift);
            PyObject *t = !(h && q2 && q) ? NULL : (tp == &PyTuple_Type) ? PyTuple_New(3) : tp->tp_alloc(tp, 3);
            if (!t) {
                Py_XDECREF(h); Py_XDECREF(q2); Py_XDECREF(q);
                Py_CLEAR(list);
                goto done;
            }
            PyTuple_SET_ITEM(t, 0, h);
            PyTuple_SET_ITEM(t, 1, q2);
            PyTuple_SET_ITEM(t, 2, q);
            UNTRACK(t);
            PyObject *old = PyList_GET_ITEM(list, (Py_ssize_t)ix[i]);
            PyList_SET_ITEM(list, (Py_ssize_t)ix[i], t);
            Py_DECREF(old);
        }
    }
done:
    PyBuffer_Release(&idx); PyBuffer_Release(&buf); PyBuffer_Release(&rows);
    return list;
}

static PyMethodDef methods[] = {
    {"sparse", sparse, METH_VARARGS,
     "sparse(n_total, index, col, coloff) -> list of n_total items: bytes of the gathered column at the kept rows' ordinals, None elsewhere"},
    {"sparse_entries", sparse_entries, METH_VARARGS,
     "sparse_entries(n_total, index, buf, rows, shift, cls=None) -> list of n_total items: (header, sequence, quality) at the kept rows' ordinals, None elsewhere"},
    {"entries_phred", entries_phred, METH_VARARGS,
     "entries_phred(buf, rows, shift, qual, qoff, array_type) -> list of (header, sequence, array('b') of decoded qualities)"},
    {"entries", entries, METH_VARARGS,
     "entries(buf, rows, shift=0, hskip=1, cls=None) -> list of (header, sequence, quality) bytes tuples, one per row of six int64 positions"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_ffq_entries",
                                    "entry tuples of a whole offset table (default entryfunc, batched)", -1, methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__ffq_entries(void) { return PyModule_Create(&moddef); }
