/* lchd_pyext.c -- CPython helper of loco_hd_amd: list[PrimitiveAtom] -> SoA buffers in native code.
 *
 * The reference converts its Python arguments in native code (PyO3 `FromPyObject` derives clone every PrimitiveAtom of both
 * lists into Rust Vecs on each call, /root/reference/src/locohd/primitive_atom.rs:4-16, src/locohd.rs:479-485).  Doing that
 * extraction in a Python loop costs ~0.5 us per attribute and dominates a from_primitives call on structures of a few
 * thousand atoms (1.5 ms of 1.8 ms); this module does it with the C API: one pass over the list that fills caller-owned
 * NumPy buffers (xyz float64 [n][3], category int32 [n], tag int32 [n]), looks categories up in the LoCoHD instance's
 * dict and interns tags into the caller's dict exactly like LoCoHD.pack does.
 *
 *     _fastpack.pack_into(prims, categories: dict, interner: dict, xyz, cat, tag) -> None
 *
 * Host-side argument conversion only: no scoring arithmetic lives here.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject *s_coords_private, *s_coords_public, *s_type, *s_tag, *s_type_private, *s_tag_private;

/* attribute `priv` (the slot of loco_hd_amd.PrimitiveAtom: no property call), else `pub` (any object with the public names) */
static PyObject *get_attr2(PyObject *p, PyObject *priv, PyObject *pub) {
    PyObject *v = PyObject_GetAttr(p, priv);
    if (v) return v;
    PyErr_Clear();
    return PyObject_GetAttr(p, pub);
}

static int get_buffer(PyObject *obj, Py_buffer *view, Py_ssize_t itemsize, Py_ssize_t need_items, const char *what) {
    if (PyObject_GetBuffer(obj, view, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) return -1;
    if (view->itemsize != itemsize || view->len != need_items * itemsize) {
        PyErr_Format(PyExc_ValueError, "%s buffer must hold %zd items of %zd bytes", what, need_items, itemsize);
        PyBuffer_Release(view);
        return -1;
    }
    return 0;
}

static PyObject *pack_into(PyObject *self, PyObject *args) {
    PyObject *prims, *categories, *interner, *o_xyz, *o_cat, *o_tag;
    if (!PyArg_ParseTuple(args, "OO!O!OOO", &prims, &PyDict_Type, &categories, &PyDict_Type, &interner, &o_xyz, &o_cat, &o_tag)) return NULL;
    PyObject *seq = PySequence_Fast(prims, "expected a sequence of PrimitiveAtom");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    Py_buffer b_xyz, b_cat, b_tag;
    if (get_buffer(o_xyz, &b_xyz, 8, 3 * n, "xyz") != 0) { Py_DECREF(seq); return NULL; }
    if (get_buffer(o_cat, &b_cat, 4, n, "cat") != 0) { PyBuffer_Release(&b_xyz); Py_DECREF(seq); return NULL; }
    if (get_buffer(o_tag, &b_tag, 4, n, "tag") != 0) { PyBuffer_Release(&b_xyz); PyBuffer_Release(&b_cat); Py_DECREF(seq); return NULL; }
    double *xyz = (double *)b_xyz.buf;
    int32_t *cat = (int32_t *)b_cat.buf, *tag = (int32_t *)b_tag.buf;
    int ok = 1;
    for (Py_ssize_t i = 0; i < n && ok; ++i) {
        PyObject *p = PySequence_Fast_GET_ITEM(seq, i); /* borrowed */
        /* coordinates: the private list of loco_hd_amd.PrimitiveAtom, or any object's public `coordinates` */
        PyObject *c = PyObject_GetAttr(p, s_coords_private);
        if (!c) {
            PyErr_Clear();
            c = PyObject_GetAttr(p, s_coords_public);
            if (!c) { ok = 0; break; }
        }
        PyObject *cs = PySequence_Fast(c, "coordinates must be a sequence of three numbers");
        Py_DECREF(c);
        if (!cs) { ok = 0; break; }
        if (PySequence_Fast_GET_SIZE(cs) != 3) {
            PyErr_Format(PyExc_ValueError, "expected a sequence of length 3 (got %zd)", PySequence_Fast_GET_SIZE(cs));
            Py_DECREF(cs);
            ok = 0;
            break;
        }
        for (int k = 0; k < 3; ++k) {
            const double v = PyFloat_AsDouble(PySequence_Fast_GET_ITEM(cs, k));
            if (v == -1.0 && PyErr_Occurred()) { ok = 0; break; }
            xyz[3 * i + k] = v;
        }
        Py_DECREF(cs);
        if (!ok) break;
        /* category: index in the LoCoHD instance's map, -1 if absent (src/locohd/pmf.rs:38-42 raises later) */
        PyObject *t = get_attr2(p, s_type_private, s_type);
        if (!t) { ok = 0; break; }
        PyObject *idx = PyDict_GetItemWithError(categories, t); /* borrowed */
        if (!idx && PyErr_Occurred()) { Py_DECREF(t); ok = 0; break; }
        if (!idx && !PyUnicode_CheckExact(t)) { /* LoCoHD._cats looks str(name) up: do the same for non-str labels */
            PyObject *ts = PyObject_Str(t);
            if (!ts) { Py_DECREF(t); ok = 0; break; }
            idx = PyDict_GetItemWithError(categories, ts);
            Py_DECREF(ts);
            if (!idx && PyErr_Occurred()) { Py_DECREF(t); ok = 0; break; }
        }
        Py_DECREF(t);
        if (idx) {
            const long v = PyLong_AsLong(idx);
            if (v == -1 && PyErr_Occurred()) { ok = 0; break; }
            cat[i] = (int32_t)v;
        } else {
            cat[i] = -1;
        }
        /* tag: interned in order of first appearance */
        PyObject *g = get_attr2(p, s_tag_private, s_tag);
        if (!g) { ok = 0; break; }
        PyObject *id = PyDict_GetItemWithError(interner, g); /* borrowed */
        if (!id && PyErr_Occurred()) { Py_DECREF(g); ok = 0; break; }
        if (id) {
            const long v = PyLong_AsLong(id);
            if (v == -1 && PyErr_Occurred()) { Py_DECREF(g); ok = 0; break; }
            tag[i] = (int32_t)v;
        } else {
            const Py_ssize_t next = PyDict_Size(interner);
            PyObject *nv = PyLong_FromSsize_t(next);
            if (!nv || PyDict_SetItem(interner, g, nv) != 0) { Py_XDECREF(nv); Py_DECREF(g); ok = 0; break; }
            Py_DECREF(nv);
            tag[i] = (int32_t)next;
        }
        Py_DECREF(g);
    }
    PyBuffer_Release(&b_xyz);
    PyBuffer_Release(&b_cat);
    PyBuffer_Release(&b_tag);
    Py_DECREF(seq);
    if (!ok) return NULL;
    Py_RETURN_NONE;
}

/* cats_into(seq, categories: dict, cat) -> None: category index of every label of a sequence (LoCoHD._cats: the seq_a /
 * seq_b arguments of from_anchors / from_dmxs / from_coords, Vec<String> in the reference), -1 for labels outside the map. */
static PyObject *cats_into(PyObject *self, PyObject *args) {
    PyObject *labels, *categories, *o_cat;
    if (!PyArg_ParseTuple(args, "OO!O", &labels, &PyDict_Type, &categories, &o_cat)) return NULL;
    PyObject *seq = PySequence_Fast(labels, "expected a sequence of category labels");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    Py_buffer b_cat;
    if (get_buffer(o_cat, &b_cat, 4, n, "cat") != 0) { Py_DECREF(seq); return NULL; }
    int32_t *cat = (int32_t *)b_cat.buf;
    int ok = 1;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *t = PySequence_Fast_GET_ITEM(seq, i); /* borrowed */
        PyObject *idx = PyDict_GetItemWithError(categories, t); /* borrowed */
        if (!idx && PyErr_Occurred()) { ok = 0; break; }
        if (!idx && !PyUnicode_CheckExact(t)) { /* str(label), as the Python loop does */
            PyObject *ts = PyObject_Str(t);
            if (!ts) { ok = 0; break; }
            idx = PyDict_GetItemWithError(categories, ts);
            Py_DECREF(ts);
            if (!idx && PyErr_Occurred()) { ok = 0; break; }
        }
        if (idx) {
            const long v = PyLong_AsLong(idx);
            if (v == -1 && PyErr_Occurred()) { ok = 0; break; }
            cat[i] = (int32_t)v;
        } else {
            cat[i] = -1;
        }
    }
    PyBuffer_Release(&b_cat);
    Py_DECREF(seq);
    if (!ok) return NULL;
    Py_RETURN_NONE;
}

/* items_tuple(seq, cls) -> tuple(seq) if seq is a list / tuple whose items are all EXACTLY of type cls (cls may be None: any
 * items), else None.  same_items(seq, tup) -> True iff seq is a list / tuple with the very same item objects as tup.
 * The pair implements the identity check of LoCoHD's packed-structure cache: the tuple keeps the items alive (an address cannot be
 * re-used by another object), so "same objects" + "no PrimitiveAtom setter ran since" means "same content". */
static PyObject *items_tuple(PyObject *self, PyObject *args) {
    PyObject *seq, *cls;
    if (!PyArg_ParseTuple(args, "OO", &seq, &cls)) return NULL;
    if (!PyList_CheckExact(seq) && !PyTuple_CheckExact(seq)) Py_RETURN_NONE;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject **items = PySequence_Fast_ITEMS(seq);
    if (cls != Py_None)
        for (Py_ssize_t i = 0; i < n; ++i)
            if ((PyObject *)Py_TYPE(items[i]) != cls) Py_RETURN_NONE;
    PyObject *tup = PyTuple_New(n);
    if (!tup) return NULL;
    for (Py_ssize_t i = 0; i < n; ++i) {
        Py_INCREF(items[i]);
        PyTuple_SET_ITEM(tup, i, items[i]);
    }
    return tup;
}
static PyObject *same_items(PyObject *self, PyObject *args) {
    PyObject *seq, *tup;
    if (!PyArg_ParseTuple(args, "OO!", &seq, &PyTuple_Type, &tup)) return NULL;
    if (!PyList_CheckExact(seq) && !PyTuple_CheckExact(seq)) Py_RETURN_FALSE;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    if (n != PyTuple_GET_SIZE(tup)) Py_RETURN_FALSE;
    PyObject **a = PySequence_Fast_ITEMS(seq), **b = PySequence_Fast_ITEMS(tup);
    for (Py_ssize_t i = 0; i < n; ++i)
        if (a[i] != b[i]) Py_RETURN_FALSE;
    Py_RETURN_TRUE;
}

static PyMethodDef methods[] = {
    {"items_tuple", items_tuple, METH_VARARGS, "items_tuple(seq, cls): tuple(seq) if every item is exactly of type cls, else None"},
    {"same_items", same_items, METH_VARARGS, "same_items(seq, tup): the very same item objects?"},
    {"cats_into", cats_into, METH_VARARGS, "cats_into(labels, categories, cat): category indices of a sequence of labels (-1: not in the map)"},
    {"pack_into", pack_into, METH_VARARGS, "pack_into(prims, categories, interner, xyz, cat, tag): fill SoA buffers from a sequence of PrimitiveAtom"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fastpack", "native list[PrimitiveAtom] -> SoA conversion", -1, methods};

PyMODINIT_FUNC PyInit__fastpack(void) {
    s_coords_private = PyUnicode_InternFromString("_coordinates");
    s_coords_public = PyUnicode_InternFromString("coordinates");
    s_type = PyUnicode_InternFromString("primitive_type");
    s_tag = PyUnicode_InternFromString("tag");
    s_type_private = PyUnicode_InternFromString("_primitive_type");
    s_tag_private = PyUnicode_InternFromString("_tag");
    if (!s_coords_private || !s_coords_public || !s_type || !s_tag || !s_type_private || !s_tag_private) return NULL;
    return PyModule_Create(&moduledef);
}
