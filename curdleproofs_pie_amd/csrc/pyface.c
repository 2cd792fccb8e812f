/* _pyface -- marshalling helper of the Python face (curdleproofs_pie_amd/py_arkworks_bls12381.py, msm_accumulator.py).
 *
 * The reference's boundary is lists of opaque G1Point / Scalar OBJECTS (curdleproofs/curdleproofs/msm_accumulator.py:6-12,
 * :37-58): before a single byte reaches the GPU, n point blobs and n integers have to be gathered out of n Python objects.
 * Done in Python that is ~0.2 us per attribute + 0.18 us per int.to_bytes: 12 ms of joins around a 0.78 ms device call at
 * n = 2^16.  This module walks the two sequences in C and writes straight into the caller's (page-locked) staging
 * buffers: one call per list.
 *
 * Nothing here computes: no field or group arithmetic, no fallback for any device path.  Without it the Python face still
 * works (pure-Python packing); with it compute_MSM's host side is memcpy-bound.
 *
 * Built by curdleproofs_pie_amd/build.py with gcc against Python.h into curdleproofs_pie_amd/_pyface.<abi>.so.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>
#include <stdint.h>
#include <string.h>

#define POINT_BYTES 144

static PyTypeObject* g_point_type = NULL;   /* G1Point: slot `_blob` = bytes(144) (host Jacobian X | Y | Z, Montgomery 2^384) or None while the value is deferred */
static PyTypeObject* g_scalar_type = NULL;  /* Scalar:  slot `_v` = int in [0, r) */
static Py_ssize_t g_point_off = -1, g_scalar_off = -1;
static Py_ssize_t g_cache_off[2] = {-1, -1};   /* G1Point slots `_a`, `_k` (normal-form caches; None until filled) */
/* the remaining G1Point slots (`_t` deferred terms, `_sg` subgroup certainty, `_seq` creation number), set to None by points_from_blobs */
#define PF_MAX_EXTRA 6
static Py_ssize_t g_extra_off[PF_MAX_EXTRA];
static int g_n_extra = 0;
static Py_ssize_t g_t_off = -1, g_sg_off = -1, g_seq_off = -1;
static PyObject* g_unforced = NULL;         /* exception: a point of the sequence has no blob yet (the caller evaluates it and retries) */
static PyObject* g_pending = NULL;          /* the Python face's list of weak references to unevaluated values (bind) */
static PyObject* g_identity_blob = NULL;
typedef int (*validate_fn)(const unsigned char*, int*);
static validate_fn g_validate = NULL;       /* cg1_validate_compressed of libcurdle_g1.so (set_native) */

/* Montgomery form of 1 (csrc/bls_consts.h H_R1): Z of a normalised point */
static const uint64_t MONT_ONE[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                                     0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};

static Py_ssize_t slot_offset(PyTypeObject* tp, const char* name) {
  PyObject* d = PyDict_GetItemString(tp->tp_dict, name);      /* borrowed */
  if (!d || Py_TYPE(d) != &PyMemberDescr_Type) return -1;
  PyMemberDef* m = ((PyMemberDescrObject*)d)->d_member;
  if (!m || m->type != T_OBJECT_EX) return -1;
  return m->offset;
}

/* bind(G1Point, Scalar[, pending_list, identity_blob]): remember the two classes and where their slots live */
static PyObject* pf_bind(PyObject* self, PyObject* args) {
  PyObject *pt, *sc, *pending = NULL, *idb = NULL;
  if (!PyArg_ParseTuple(args, "OO|OO", &pt, &sc, &pending, &idb)) return NULL;
  if (!PyType_Check(pt) || !PyType_Check(sc)) { PyErr_SetString(PyExc_TypeError, "bind(G1Point, Scalar) expects two classes"); return NULL; }
  Py_ssize_t po = slot_offset((PyTypeObject*)pt, "_blob"), so = slot_offset((PyTypeObject*)sc, "_v");
  if (po < 0 || so < 0) { PyErr_SetString(PyExc_TypeError, "G1Point._blob / Scalar._v are not __slots__ members"); return NULL; }
  if (pending && pending != Py_None && !PyList_CheckExact(pending)) { PyErr_SetString(PyExc_TypeError, "pending must be a list"); return NULL; }
  if (idb && idb != Py_None && (!PyBytes_CheckExact(idb) || PyBytes_GET_SIZE(idb) != 144)) { PyErr_SetString(PyExc_TypeError, "identity blob must be 144 bytes"); return NULL; }
  Py_INCREF(pt); Py_INCREF(sc);
  Py_XDECREF(g_point_type); Py_XDECREF(g_scalar_type);
  g_point_type = (PyTypeObject*)pt; g_scalar_type = (PyTypeObject*)sc;
  g_point_off = po; g_scalar_off = so;
  g_cache_off[0] = slot_offset((PyTypeObject*)pt, "_a");
  g_cache_off[1] = slot_offset((PyTypeObject*)pt, "_k");
  g_t_off = slot_offset((PyTypeObject*)pt, "_t");
  g_sg_off = slot_offset((PyTypeObject*)pt, "_sg");
  g_seq_off = slot_offset((PyTypeObject*)pt, "_seq");
  g_n_extra = 0;
  if (g_t_off >= 0) g_extra_off[g_n_extra++] = g_t_off;
  if (g_sg_off >= 0) g_extra_off[g_n_extra++] = g_sg_off;
  if (g_seq_off >= 0) g_extra_off[g_n_extra++] = g_seq_off;
  Py_XDECREF(g_pending); g_pending = NULL;
  if (pending && pending != Py_None) { Py_INCREF(pending); g_pending = pending; }
  Py_XDECREF(g_identity_blob); g_identity_blob = NULL;
  if (idb && idb != Py_None) { Py_INCREF(idb); g_identity_blob = idb; }
  Py_RETURN_NONE;
}

/* set_native(addr of cg1_validate_compressed) */
static PyObject* pf_set_native(PyObject* self, PyObject* args) {
  unsigned long long a;
  if (!PyArg_ParseTuple(args, "K", &a)) return NULL;
  g_validate = (validate_fn)(uintptr_t)a;
  Py_RETURN_NONE;
}

static inline PyObject* slot_get(PyObject* o, Py_ssize_t off) { return *(PyObject**)((char*)o + off); }

/* pack_points(seq, dst_addr, capacity_points) -> (n, all_normalised)
 * Copies the 144-byte blob of every G1Point of `seq` (list or tuple) to dst_addr + 144 i.  all_normalised: every blob has
 * Z == 1 (Montgomery) or Z == 0 (identity), i.e. X, Y already are the affine coordinates and the device needs no inversion. */
/* The walks below chase two pointers per element (list item -> object -> bytes / int): with a million objects every hop is a cache
 * miss.  Each loop therefore touches element i + PF_FAR's object header and element i + PF_NEAR's payload ahead of time (the object of
 * i + PF_NEAR was fetched PF_FAR - PF_NEAR iterations earlier, so reading its slot is cheap).  Prefetches of wrong addresses are harmless;
 * the type of a look-ahead element is checked before its slot is read. */
#define PF_FAR 24
#define PF_NEAR 8
static inline void pf_touch(const void* p) { __builtin_prefetch(p, 0, 1); }

static PyObject* pf_pack_points(PyObject* self, PyObject* args) {
  PyObject* seq; unsigned long long addr; Py_ssize_t cap;
  if (!PyArg_ParseTuple(args, "OKn", &seq, &addr, &cap)) return NULL;
  if (!g_point_type) { PyErr_SetString(PyExc_RuntimeError, "_pyface.bind() has not run"); return NULL; }
  PyObject* fast = PySequence_Fast(seq, "pack_points expects a sequence of G1Point");
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  if (n > cap) { Py_DECREF(fast); PyErr_SetString(PyExc_ValueError, "staging buffer too small"); return NULL; }
  PyObject** items = PySequence_Fast_ITEMS(fast);
  uint8_t* dst = (uint8_t*)(uintptr_t)addr;
  int normalised = 1;
  for (Py_ssize_t i = 0; i < n; ++i) {
    if (i + PF_FAR < n) pf_touch(items[i + PF_FAR]);
    if (i + PF_NEAR < n && Py_TYPE(items[i + PF_NEAR]) == g_point_type) {
      const char* nb = (const char*)slot_get(items[i + PF_NEAR], g_point_off);
      if (nb) { pf_touch(nb); pf_touch(nb + 64); pf_touch(nb + 128); }
    }
    PyObject* o = items[i];
    if (Py_TYPE(o) != g_point_type) { Py_DECREF(fast); PyErr_Format(PyExc_TypeError, "element %zd is not a G1Point", i); return NULL; }
    PyObject* b = slot_get(o, g_point_off);
    if (b == Py_None) { Py_DECREF(fast); PyErr_Format(g_unforced, "element %zd is a deferred value", i); return NULL; }
    if (!b || !PyBytes_CheckExact(b) || PyBytes_GET_SIZE(b) != POINT_BYTES) { Py_DECREF(fast); PyErr_Format(PyExc_TypeError, "element %zd holds no 144-byte blob", i); return NULL; }
    const char* src = PyBytes_AS_STRING(b);
    memcpy(dst + (size_t)POINT_BYTES * (size_t)i, src, POINT_BYTES);
    if (normalised) {
      uint64_t z[6];
      memcpy(z, src + 96, 48);
      const uint64_t nz = z[0] | z[1] | z[2] | z[3] | z[4] | z[5];
      if (nz && memcmp(z, MONT_ONE, 48) != 0) normalised = 0;
    }
  }
  Py_DECREF(fast);
  return Py_BuildValue("ni", n, normalised);
}

/* pack_affine(seq, dst_addr, capacity_points) -> n
 * Copies the cached affine96 record (slot `_a`, filled by py_arkworks_bls12381.ensure_normalised) of every G1Point to dst_addr + 96 i. */
static PyObject* pf_pack_affine(PyObject* self, PyObject* args) {
  PyObject* seq; unsigned long long addr; Py_ssize_t cap;
  if (!PyArg_ParseTuple(args, "OKn", &seq, &addr, &cap)) return NULL;
  if (!g_point_type || g_cache_off[0] < 0) { PyErr_SetString(PyExc_RuntimeError, "_pyface.bind() has not run"); return NULL; }
  PyObject* fast = PySequence_Fast(seq, "pack_affine expects a sequence of G1Point");
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  if (n > cap) { Py_DECREF(fast); PyErr_SetString(PyExc_ValueError, "staging buffer too small"); return NULL; }
  PyObject** items = PySequence_Fast_ITEMS(fast);
  uint8_t* dst = (uint8_t*)(uintptr_t)addr;
  for (Py_ssize_t i = 0; i < n; ++i) {
    if (i + PF_FAR < n) pf_touch(items[i + PF_FAR]);
    if (i + PF_NEAR < n && Py_TYPE(items[i + PF_NEAR]) == g_point_type) {
      const char* nb = (const char*)slot_get(items[i + PF_NEAR], g_cache_off[0]);
      if (nb) { pf_touch(nb); pf_touch(nb + 64); }
    }
    PyObject* o = items[i];
    if (Py_TYPE(o) != g_point_type) { Py_DECREF(fast); PyErr_Format(PyExc_TypeError, "element %zd is not a G1Point", i); return NULL; }
    PyObject* b = slot_get(o, g_cache_off[0]);
    if (!b || !PyBytes_CheckExact(b) || PyBytes_GET_SIZE(b) != 96) { Py_DECREF(fast); PyErr_Format(PyExc_ValueError, "element %zd has not been normalised", i); return NULL; }
    memcpy(dst + 96 * (size_t)i, PyBytes_AS_STRING(b), 96);
  }
  Py_DECREF(fast);
  return PyLong_FromSsize_t(n);
}

/* A non-negative Python int below 2^256 as 32 little-endian bytes, read straight from its 30-bit digits (CPython's own
 * _PyLong_AsByteArray walks the digits a byte at a time: 25 of the 33 ns an element costs).  Returns 0 when the value does not qualify
 * (negative, too long, other digit width, big-endian host): the caller then lets _PyLong_AsByteArray decide and raise. */
static inline int long_to_le32(PyObject* v, uint8_t* out) {
#if PY_VERSION_HEX < 0x030C0000 && PYLONG_BITS_IN_DIGIT == 30 && defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__   /* (3.12 moved the digit count into lv_tag) */
  const PyLongObject* l = (const PyLongObject*)v;
  const Py_ssize_t sz = Py_SIZE(l);
  if (sz < 0 || sz > 9) return 0;
  uint64_t w[6] = {0, 0, 0, 0, 0, 0};
  for (Py_ssize_t k = 0; k < sz; ++k) {
    const uint64_t d = (uint64_t)l->ob_digit[k];
    const unsigned bit = 30u * (unsigned)k, sh = bit & 63u;
    w[bit >> 6] |= d << sh;
    if (sh > 34u) w[(bit >> 6) + 1] |= d >> (64u - sh);
  }
  if (w[4] | w[5]) return 0;
  memcpy(out, w, 32);
  return 1;
#else
  (void)v; (void)out;
  return 0;
#endif
}

/* the interpreter's own conversion (any digit layout); -1 with OverflowError set for a negative value or one of 2^256 and more */
static inline int pf_long_as_le32(PyObject* v, uint8_t* out) {
#if PY_VERSION_HEX >= 0x030D0000
  return _PyLong_AsByteArray((PyLongObject*)v, out, 32, 1, 0, 1);
#else
  return _PyLong_AsByteArray((PyLongObject*)v, out, 32, 1, 0);
#endif
}

/* pack_scalars(seq, dst_addr, capacity) -> n
 * Writes int(s) of every Scalar of `seq` as 32 little-endian bytes to dst_addr + 32 i (Scalar.to_le_bytes, one call). */
static PyObject* pf_pack_scalars(PyObject* self, PyObject* args) {
  PyObject* seq; unsigned long long addr; Py_ssize_t cap;
  if (!PyArg_ParseTuple(args, "OKn", &seq, &addr, &cap)) return NULL;
  if (!g_scalar_type) { PyErr_SetString(PyExc_RuntimeError, "_pyface.bind() has not run"); return NULL; }
  PyObject* fast = PySequence_Fast(seq, "pack_scalars expects a sequence of Scalar");
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  if (n > cap) { Py_DECREF(fast); PyErr_SetString(PyExc_ValueError, "staging buffer too small"); return NULL; }
  PyObject** items = PySequence_Fast_ITEMS(fast);
  uint8_t* dst = (uint8_t*)(uintptr_t)addr;
  for (Py_ssize_t i = 0; i < n; ++i) {
    if (i + PF_FAR < n) pf_touch(items[i + PF_FAR]);
    if (i + PF_NEAR < n && Py_TYPE(items[i + PF_NEAR]) == g_scalar_type) {
      const void* nv = slot_get(items[i + PF_NEAR], g_scalar_off);
      if (nv) pf_touch(nv);
    }
    PyObject* o = items[i];
    PyObject* v;
    if (Py_TYPE(o) == g_scalar_type) v = slot_get(o, g_scalar_off);
    else if (PyLong_CheckExact(o)) v = o;                       /* plain ints are accepted: the accumulator keeps merged scalars as ints */
    else { Py_DECREF(fast); PyErr_Format(PyExc_TypeError, "element %zd is not a Scalar", i); return NULL; }
    if (!v || !PyLong_Check(v)) { Py_DECREF(fast); PyErr_Format(PyExc_TypeError, "element %zd holds no integer", i); return NULL; }
    if (!long_to_le32(v, dst + 32 * (size_t)i) && pf_long_as_le32(v, dst + 32 * (size_t)i) < 0) { Py_DECREF(fast); return NULL; }   /* OverflowError: negative or >= 2^256 */
  }
  Py_DECREF(fast);
  return PyLong_FromSsize_t(n);
}

/* points_from_blobs(data, n) -> [G1Point, ...]: n objects over consecutive 144-byte blobs of a bytes-like object */
static PyObject* pf_points_from_blobs(PyObject* self, PyObject* args) {
  Py_buffer view; Py_ssize_t n;
  if (!PyArg_ParseTuple(args, "y*n", &view, &n)) return NULL;
  if (!g_point_type) { PyBuffer_Release(&view); PyErr_SetString(PyExc_RuntimeError, "_pyface.bind() has not run"); return NULL; }
  if (n < 0 || view.len < n * POINT_BYTES) { PyBuffer_Release(&view); PyErr_SetString(PyExc_ValueError, "buffer shorter than n blobs"); return NULL; }
  PyObject* out = PyList_New(n);
  if (!out) { PyBuffer_Release(&view); return NULL; }
  const char* src = (const char*)view.buf;
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject* o = g_point_type->tp_alloc(g_point_type, 0);
    PyObject* b = o ? PyBytes_FromStringAndSize(src + (size_t)POINT_BYTES * (size_t)i, POINT_BYTES) : NULL;
    if (!b) { Py_XDECREF(o); Py_DECREF(out); PyBuffer_Release(&view); return NULL; }
    *(PyObject**)((char*)o + g_point_off) = b;
    for (int c = 0; c < 2; ++c)
      if (g_cache_off[c] >= 0) { Py_INCREF(Py_None); *(PyObject**)((char*)o + g_cache_off[c]) = Py_None; }
    for (int c = 0; c < g_n_extra; ++c) { Py_INCREF(Py_None); *(PyObject**)((char*)o + g_extra_off[c]) = Py_None; }
    PyList_SET_ITEM(out, i, o);
  }
  PyBuffer_Release(&view);
  return out;
}

/* ident(seq) -> (n, fingerprint): a 64-bit mix of the element IDENTITIES (addresses) of a list / tuple.  G1Point objects are
 * immutable, so two sequences of the very same objects hold the same points: the key of the resident-vector cache. */
static PyObject* pf_ident(PyObject* self, PyObject* args) {
  PyObject* seq;
  if (!PyArg_ParseTuple(args, "O", &seq)) return NULL;
  PyObject* fast = PySequence_Fast(seq, "ident expects a sequence");
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  PyObject** items = PySequence_Fast_ITEMS(fast);
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
  for (Py_ssize_t i = 0; i < n; ++i) {
    h ^= (uint64_t)(uintptr_t)items[i];
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 29;
  }
  Py_DECREF(fast);
  return Py_BuildValue("nK", n, (unsigned long long)h);
}

/* same_items(seq, tup) -> bool: the two sequences hold the very same objects in the same order */
static PyObject* pf_same_items(PyObject* self, PyObject* args) {
  PyObject *a, *b;
  if (!PyArg_ParseTuple(args, "OO", &a, &b)) return NULL;
  PyObject* fa = PySequence_Fast(a, "same_items expects sequences");
  if (!fa) return NULL;
  PyObject* fb = PySequence_Fast(b, "same_items expects sequences");
  if (!fb) { Py_DECREF(fa); return NULL; }
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fa);
  int same = n == PySequence_Fast_GET_SIZE(fb) &&
             (n == 0 || memcmp(PySequence_Fast_ITEMS(fa), PySequence_Fast_ITEMS(fb), (size_t)n * sizeof(PyObject*)) == 0);
  Py_DECREF(fa); Py_DECREF(fb);
  if (same) Py_RETURN_TRUE;
  Py_RETURN_FALSE;
}

static inline void slot_put(PyObject* o, Py_ssize_t off, PyObject* v) {
  if (off < 0) return;
  Py_INCREF(v);
  *(PyObject**)((char*)o + off) = v;
}

/* mk(blob, a, k, t, sg, seq) -> G1Point with exactly these slot values; a value with terms (t is not None) is entered in the pending list */
static PyObject* pf_mk(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 6) { PyErr_SetString(PyExc_TypeError, "mk(blob, a, k, t, sg, seq)"); return NULL; }
  if (!g_point_type || g_t_off < 0 || g_sg_off < 0 || g_seq_off < 0) { PyErr_SetString(PyExc_RuntimeError, "_pyface.bind() has not run"); return NULL; }
  PyObject* o = g_point_type->tp_alloc(g_point_type, 0);
  if (!o) return NULL;
  slot_put(o, g_point_off, args[0]);
  slot_put(o, g_cache_off[0], args[1]);
  slot_put(o, g_cache_off[1], args[2]);
  slot_put(o, g_t_off, args[3]);
  slot_put(o, g_sg_off, args[4]);
  slot_put(o, g_seq_off, args[5]);
  if (args[3] != Py_None && g_pending) {
    PyObject* r = PyWeakref_NewRef(o, NULL);
    if (!r || PyList_Append(g_pending, r) < 0) { Py_XDECREF(r); Py_DECREF(o); return NULL; }
    Py_DECREF(r);
  }
  return o;
}

/* decode_lazy(data: bytes of 48) -> G1Point: the encoding validated now (flags, x < p, on the curve: libcurdle_g1's cg1_validate_compressed,
 * a Jacobi symbol instead of the square root), y left for later: the object holds the 48 bytes as its `_k` and no blob.  ValueError as
 * G1Point.from_compressed_bytes_unchecked raises it. */
static PyObject* pf_decode_lazy(PyObject* self, PyObject* data) {
  if (!g_point_type || !g_validate || !g_identity_blob || g_t_off < 0) { PyErr_SetString(PyExc_RuntimeError, "_pyface.bind() / set_native() have not run"); return NULL; }
  PyObject* owned = NULL;
  if (!PyBytes_CheckExact(data)) {
    owned = PyObject_Bytes(data);
    if (!owned) return NULL;
    data = owned;
  }
  if (PyBytes_GET_SIZE(data) != 48) { Py_XDECREF(owned); PyErr_SetString(PyExc_ValueError, "Err From Rust: serialised data seems to be invalid (need 48 bytes)"); return NULL; }
  int inf = 0;
  const int rc = g_validate((const unsigned char*)PyBytes_AS_STRING(data), &inf);
  if (rc != 0) { Py_XDECREF(owned); PyErr_Format(PyExc_ValueError, "Err From Rust: serialised data seems to be invalid (code %d)", rc); return NULL; }
  PyObject* o = g_point_type->tp_alloc(g_point_type, 0);
  if (!o) { Py_XDECREF(owned); return NULL; }
  slot_put(o, g_point_off, inf ? g_identity_blob : Py_None);
  slot_put(o, g_cache_off[0], Py_None);
  slot_put(o, g_cache_off[1], inf ? Py_None : data);       /* a finite point decodes from exactly one encoding: these bytes ARE its compression */
  slot_put(o, g_t_off, Py_None);
  slot_put(o, g_sg_off, inf ? Py_True : Py_None);
  slot_put(o, g_seq_off, Py_None);
  Py_XDECREF(owned);
  return o;
}

static PyMethodDef methods[] = {
    {"mk", (PyCFunction)(void (*)(void))pf_mk, METH_FASTCALL, "mk(blob, a, k, t, sg, seq) -> G1Point"},
    {"decode_lazy", pf_decode_lazy, METH_O, "decode_lazy(data48) -> G1Point (validated, y deferred)"},
    {"set_native", pf_set_native, METH_VARARGS, "set_native(address of cg1_validate_compressed)"},
    {"ident", pf_ident, METH_VARARGS, "ident(seq) -> (n, fingerprint of the element identities)"},
    {"same_items", pf_same_items, METH_VARARGS, "same_items(a, b) -> bool"},
    {"bind", pf_bind, METH_VARARGS, "bind(G1Point, Scalar)"},
    {"pack_points", pf_pack_points, METH_VARARGS, "pack_points(seq, dst_addr, capacity) -> (n, all_normalised)"},
    {"pack_affine", pf_pack_affine, METH_VARARGS, "pack_affine(seq, dst_addr, capacity) -> n"},
    {"pack_scalars", pf_pack_scalars, METH_VARARGS, "pack_scalars(seq, dst_addr, capacity) -> n"},
    {"points_from_blobs", pf_points_from_blobs, METH_VARARGS, "points_from_blobs(data, n) -> list of G1Point"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pyface", "marshalling helper of the curdleproofs_pie_amd Python face", -1, methods};

PyMODINIT_FUNC PyInit__pyface(void) {
  PyObject* m = PyModule_Create(&moddef);
  if (!m) return NULL;
  g_unforced = PyErr_NewException("_pyface.Unforced", PyExc_LookupError, NULL);
  if (!g_unforced) { Py_DECREF(m); return NULL; }
  Py_INCREF(g_unforced);
  if (PyModule_AddObject(m, "Unforced", g_unforced) < 0) { Py_DECREF(g_unforced); Py_DECREF(m); return NULL; }
  return m;
}
