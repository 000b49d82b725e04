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
#include <structmember.h>

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


/* pack_atoms(prims, cls, type_map, xyz, cat, tag) -> bool
 *
 * The fast form of pack_into for lists / tuples whose items are EXACTLY loco_hd_amd.PrimitiveAtom: the class interns its two strings
 * when an atom is constructed or changed (slots `_pid`: id of the primitive type, `_tid`: id of the tag, both from process-wide
 * tables), so packing a list is a gather of three doubles and two integers per atom -- slots read at their offsets, no attribute
 * look-up, no hashing (the reference's PyO3 extraction clones two Strings per atom instead, primitive_atom.rs:4-16).
 * type_map: int32 buffer, primitive-type id -> category index of the calling LoCoHD instance (-1: not in its map).
 * Returns False -- nothing usable written -- when the sequence is not a list / tuple of exactly `cls`, or an atom lacks its ids
 * (the caller then takes pack_into). */
static Py_ssize_t slot_offset(PyObject *cls, const char *name) {
    PyObject *d = PyObject_GetAttrString(cls, name);
    if (!d) { PyErr_Clear(); return -1; }
    Py_ssize_t off = -1;
    if (Py_TYPE(d) == &PyMemberDescr_Type) {
        PyMemberDef *m = ((PyMemberDescrObject *)d)->d_member;
        if (m && m->type == T_OBJECT_EX) off = m->offset;
    }
    Py_DECREF(d);
    return off;
}

static PyObject *pack_atoms(PyObject *self, PyObject *args) {
    PyObject *prims, *cls, *o_map, *o_xyz, *o_cat, *o_tag;
    if (!PyArg_ParseTuple(args, "OOOOOO", &prims, &cls, &o_map, &o_xyz, &o_cat, &o_tag)) return NULL;
    if (!PyList_CheckExact(prims) && !PyTuple_CheckExact(prims)) Py_RETURN_FALSE;
    if (!PyType_Check(cls)) Py_RETURN_FALSE;
    const Py_ssize_t oc = slot_offset(cls, "_coordinates"), op = slot_offset(cls, "_pid"), ot = slot_offset(cls, "_tid");
    if (oc < 0 || op < 0 || ot < 0) Py_RETURN_FALSE;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(prims);
    PyObject **items = PySequence_Fast_ITEMS(prims);
    Py_buffer b_map, b_xyz, b_cat, b_tag;
    if (PyObject_GetBuffer(o_map, &b_map, PyBUF_C_CONTIGUOUS) != 0) return NULL;
    if (b_map.itemsize != 4) { PyBuffer_Release(&b_map); PyErr_SetString(PyExc_ValueError, "type_map must be int32"); return NULL; }
    if (get_buffer(o_xyz, &b_xyz, 8, 3 * n, "xyz") != 0) { PyBuffer_Release(&b_map); return NULL; }
    if (get_buffer(o_cat, &b_cat, 4, n, "cat") != 0) { PyBuffer_Release(&b_map); PyBuffer_Release(&b_xyz); return NULL; }
    if (get_buffer(o_tag, &b_tag, 4, n, "tag") != 0) { PyBuffer_Release(&b_map); PyBuffer_Release(&b_xyz); PyBuffer_Release(&b_cat); return NULL; }
    const int32_t *map = (const int32_t *)b_map.buf;
    const Py_ssize_t n_map = b_map.len / 4;
    double *xyz = (double *)b_xyz.buf;
    int32_t *cat = (int32_t *)b_cat.buf, *tag = (int32_t *)b_tag.buf;
    int ok = 1, err = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *p = items[i];
        if ((PyObject *)Py_TYPE(p) != cls) { ok = 0; break; }
        PyObject *c = *(PyObject **)((char *)p + oc), *pid = *(PyObject **)((char *)p + op), *tid = *(PyObject **)((char *)p + ot);
        if (!c || !pid || !tid || !PyLong_CheckExact(pid) || !PyLong_CheckExact(tid)) { ok = 0; break; }
        if (PyList_CheckExact(c) && PyList_GET_SIZE(c) == 3 && PyFloat_CheckExact(PyList_GET_ITEM(c, 0)) &&
            PyFloat_CheckExact(PyList_GET_ITEM(c, 1)) && PyFloat_CheckExact(PyList_GET_ITEM(c, 2))) {
            xyz[3 * i] = PyFloat_AS_DOUBLE(PyList_GET_ITEM(c, 0));
            xyz[3 * i + 1] = PyFloat_AS_DOUBLE(PyList_GET_ITEM(c, 1));
            xyz[3 * i + 2] = PyFloat_AS_DOUBLE(PyList_GET_ITEM(c, 2));
        } else {
            ok = 0;  /* (coordinates that are not the list of three floats the constructor and the setter store: the general path) */
            break;
        }
        const long pv = PyLong_AsLong(pid), tv = PyLong_AsLong(tid);
        if ((pv == -1 || tv == -1) && PyErr_Occurred()) { err = 1; break; }
        if (pv < 0 || tv < 0 || pv >= n_map) { ok = 0; break; }
        cat[i] = map[pv];
        tag[i] = (int32_t)tv;
    }
    PyBuffer_Release(&b_map);
    PyBuffer_Release(&b_xyz);
    PyBuffer_Release(&b_cat);
    PyBuffer_Release(&b_tag);
    if (err) return NULL;
    if (ok) Py_RETURN_TRUE;
    Py_RETURN_FALSE;
}

/* pairs_into(anchor_pairs, arr) -> None | list
 *
 * AnchorPairSpecifier (src/locohd.rs:34-40) in native code: a list / tuple of 2-tuples (-> None) or of 3-tuples (-> the list of
 * their third items, the weight-function keys; an EMPTY sequence is the 3-tuple variant, as the derive order makes it);
 * arr: int64 buffer [n][2].  Mixed lengths raise TypeError, a negative index OverflowError (usize), like the Python loop. */
static PyObject *pairs_into(PyObject *self, PyObject *args) {
    PyObject *ap, *o_arr;
    if (!PyArg_ParseTuple(args, "OO", &ap, &o_arr)) return NULL;
    PyObject *seq = PySequence_Fast(ap, "expected a sequence of anchor pairs");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    Py_buffer b;
    if (get_buffer(o_arr, &b, 8, 2 * n, "anchor") != 0) { Py_DECREF(seq); return NULL; }
    int64_t *out = (int64_t *)b.buf;
    PyObject *keys = NULL;
    int ok = 1;
    Py_ssize_t width = 0;  /* 0: not known yet */
    if (n == 0) { keys = PyList_New(0); ok = keys != NULL; }
    for (Py_ssize_t i = 0; i < n && ok; ++i) {
        PyObject *t = PySequence_Fast(PySequence_Fast_GET_ITEM(seq, i), "an anchor pair must be a tuple");
        if (!t) { ok = 0; break; }
        const Py_ssize_t w = PySequence_Fast_GET_SIZE(t);
        if (width == 0) {
            width = w;
            if (w == 3) { keys = PyList_New(n); if (!keys) { Py_DECREF(t); ok = 0; break; } }
        }
        if ((w != 2 && w != 3) || w != width) {
            PyErr_SetString(PyExc_TypeError, "failed to extract enum AnchorPairSpecifier ('WithWeightFunctionKey | WithoutWeightFunctionKey')");
            Py_DECREF(t);
            ok = 0;
            break;
        }
        for (int k = 0; k < 2 && ok; ++k) {
            PyObject *v = PyNumber_Index(PySequence_Fast_GET_ITEM(t, k));
            if (!v) { ok = 0; break; }
            const long long x = PyLong_AsLongLong(v);
            Py_DECREF(v);
            if (x == -1 && PyErr_Occurred()) { ok = 0; break; }
            if (x < 0) { PyErr_SetString(PyExc_OverflowError, "can't convert negative int to unsigned"); ok = 0; break; }
            out[2 * i + k] = (int64_t)x;
        }
        if (ok && w == 3) {
            PyObject *ks = PyObject_Str(PySequence_Fast_GET_ITEM(t, 2));
            if (!ks) ok = 0; else PyList_SET_ITEM(keys, i, ks);
        }
        Py_DECREF(t);
    }
    PyBuffer_Release(&b);
    Py_DECREF(seq);
    if (!ok) { Py_XDECREF(keys); return NULL; }
    if (keys) return keys;
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
    {"pack_atoms", pack_atoms, METH_VARARGS, "pack_atoms(prims, cls, type_map, xyz, cat, tag): SoA buffers from a list of exactly-PrimitiveAtom objects through their interned ids; False if the list does not qualify"},
    {"pairs_into", pairs_into, METH_VARARGS, "pairs_into(anchor_pairs, arr): [n][2] int64 anchors; returns None (2-tuples) or the list of weight-function keys (3-tuples)"},
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
