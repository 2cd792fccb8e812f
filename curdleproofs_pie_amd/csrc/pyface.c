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
#include <stdlib.h>
#include <unistd.h>

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

/* ---- the walks, over a range of a sequence's items; no call into the interpreter, no reference count touched: a helper thread may run
 * them while the calling thread holds the GIL (the objects are immutable and nobody else can run Python code meanwhile).
 * Return -1 when done, else the index of the first element that cannot be handled here (wrong type, no blob, an int the digit reader does
 * not take): the caller looks at that element again under the GIL and raises what it deserves. */
static Py_ssize_t walk_points(PyObject** items, Py_ssize_t lo, Py_ssize_t hi, Py_ssize_t n, uint8_t* dst, int* all_normalised) {
  int normalised = 1;
  for (Py_ssize_t i = lo; i < hi; ++i) {
    if (i + PF_FAR < n) pf_touch(items[i + PF_FAR]);
    if (i + PF_NEAR < n && Py_TYPE(items[i + PF_NEAR]) == g_point_type) {
      const char* nb = (const char*)slot_get(items[i + PF_NEAR], g_point_off);
      if (nb) { pf_touch(nb); pf_touch(nb + 64); pf_touch(nb + 128); }
    }
    PyObject* o = items[i];
    if (Py_TYPE(o) != g_point_type) { *all_normalised = 0; return i; }
    PyObject* b = slot_get(o, g_point_off);
    if (!b || !PyBytes_CheckExact(b) || PyBytes_GET_SIZE(b) != POINT_BYTES) { *all_normalised = 0; return i; }
    const char* src = PyBytes_AS_STRING(b);
    memcpy(dst + (size_t)POINT_BYTES * (size_t)(i - lo), src, POINT_BYTES);
    if (normalised) {
      uint64_t z[6];
      memcpy(z, src + 96, 48);
      const uint64_t nz = z[0] | z[1] | z[2] | z[3] | z[4] | z[5];
      if (nz && memcmp(z, MONT_ONE, 48) != 0) normalised = 0;
    }
  }
  *all_normalised = normalised;
  return -1;
}

static inline int long_to_le32(PyObject* v, uint8_t* out);

static Py_ssize_t walk_scalars(PyObject** items, Py_ssize_t lo, Py_ssize_t hi, Py_ssize_t n, uint8_t* dst) {
  for (Py_ssize_t i = lo; i < hi; ++i) {
    if (i + PF_FAR < n) pf_touch(items[i + PF_FAR]);
    if (i + PF_NEAR < n && Py_TYPE(items[i + PF_NEAR]) == g_scalar_type) {
      const void* nv = slot_get(items[i + PF_NEAR], g_scalar_off);
      if (nv) pf_touch(nv);
    }
    PyObject* o = items[i];
    PyObject* v;
    if (Py_TYPE(o) == g_scalar_type) v = slot_get(o, g_scalar_off);
    else if (PyLong_CheckExact(o)) v = o;                       /* plain ints are accepted: the accumulator keeps merged scalars as ints */
    else return i;
    if (!v || !PyLong_Check(v)) return i;
    if (!long_to_le32(v, dst + 32 * (size_t)(i - lo))) return i;
  }
  return -1;
}

/* ---- helper threads for the walks over long sequences (2^20 objects: two pointer chases per element, every hop a cache miss -- one
 * thread spends ~11 ns per element waiting; several threads wait side by side).  A small persistent crew, started at the first long walk;
 * the calling thread takes a share too. */
#include <pthread.h>
#define PF_MAX_THREADS 8
#define PF_MT_MIN 16384                    /* shorter ranges are walked by the caller alone */
typedef struct {
  int kind;                                /* 0 points, 1 scalars */
  PyObject** items; Py_ssize_t lo, hi, n; uint8_t* dst; size_t rec;
  Py_ssize_t err[PF_MAX_THREADS]; int normalised[PF_MAX_THREADS];
  int parts;
} pf_job;
static struct {
  pthread_mutex_t m; pthread_cond_t work, done;
  pthread_t th[PF_MAX_THREADS];
  int started, parts, pending; unsigned long generation;
  pf_job* job;
} g_crew = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, 0, 0, NULL};

static void pf_run_part(pf_job* j, int part) {
  const Py_ssize_t len = j->hi - j->lo, per = (len + j->parts - 1) / j->parts;
  Py_ssize_t a = j->lo + per * part, b = a + per;
  if (a > j->hi) a = j->hi;
  if (b > j->hi) b = j->hi;
  uint8_t* dst = j->dst + j->rec * (size_t)(a - j->lo);
  j->normalised[part] = 1;
  if (a >= b) { j->err[part] = -1; return; }
  j->err[part] = j->kind == 0 ? walk_points(j->items, a, b, j->n, dst, &j->normalised[part]) : walk_scalars(j->items, a, b, j->n, dst);
}

static void* pf_crew_main(void* arg) {
  const int me = (int)(intptr_t)arg;       /* part me + 1 (the caller runs part 0) */
  unsigned long seen = 0;
  for (;;) {
    /* a walk over 2^20 objects arrives as 2 x 16 slices back to back: look out for the next one for a few microseconds before going to sleep */
    for (int spin = 0; spin < 4000 && __atomic_load_n(&g_crew.generation, __ATOMIC_ACQUIRE) == seen; ++spin) __builtin_ia32_pause();
    pthread_mutex_lock(&g_crew.m);
    while (__atomic_load_n(&g_crew.generation, __ATOMIC_ACQUIRE) == seen) pthread_cond_wait(&g_crew.work, &g_crew.m);
    seen = __atomic_load_n(&g_crew.generation, __ATOMIC_ACQUIRE);
    pf_job* j = g_crew.job;
    const int parts = g_crew.parts;        /* read under the lock: the job lives on the caller's stack only while its parts are pending */
    pthread_mutex_unlock(&g_crew.m);
    if (j && me + 1 < parts) {
      pf_run_part(j, me + 1);
      pthread_mutex_lock(&g_crew.m);
      if (__atomic_sub_fetch(&g_crew.pending, 1, __ATOMIC_ACQ_REL) == 0) pthread_cond_signal(&g_crew.done);
      pthread_mutex_unlock(&g_crew.m);
    }
  }
  return NULL;
}

static int g_walk_threads = -1;            /* -1: not decided yet; set_threads(n) overrides (1 = never use helpers) */

/* threads do not survive fork(): a forked child starts a crew of its own at its first long walk */
static void pf_crew_after_fork(void) {
  pthread_mutex_init(&g_crew.m, NULL);
  pthread_cond_init(&g_crew.work, NULL);
  pthread_cond_init(&g_crew.done, NULL);
  g_crew.started = 0; g_crew.parts = 0; g_crew.pending = 0; g_crew.job = NULL;
}

static int pf_crew_size(void) {
  if (g_walk_threads < 0) {
    long c = sysconf(_SC_NPROCESSORS_ONLN);
    const char* e = getenv("CURDLE_G1_WALK_THREADS");
    if (e && atol(e) >= 1) c = atol(e);
    if (c > PF_MAX_THREADS) c = PF_MAX_THREADS;
    if (c < 1) c = 1;
    g_walk_threads = (int)c;
  }
  return g_walk_threads;
}

/* walk [lo, hi) with the crew; returns the lowest failing index or -1; *all_normalised as walk_points */
static Py_ssize_t pf_walk(int kind, PyObject** items, Py_ssize_t lo, Py_ssize_t hi, Py_ssize_t n, uint8_t* dst, int* all_normalised) {
  pf_job j;
  j.kind = kind; j.items = items; j.lo = lo; j.hi = hi; j.n = n; j.dst = dst; j.rec = kind == 0 ? POINT_BYTES : 32;
  int parts = pf_crew_size();
  if (hi - lo < PF_MT_MIN) parts = 1;
  if (parts > 1) {
    pthread_mutex_lock(&g_crew.m);
    while (g_crew.started < parts - 1) {
      if (pthread_create(&g_crew.th[g_crew.started], NULL, pf_crew_main, (void*)(intptr_t)g_crew.started) != 0) break;
      pthread_detach(g_crew.th[g_crew.started]);
      ++g_crew.started;
    }
    if (parts > g_crew.started + 1) parts = g_crew.started + 1;
    pthread_mutex_unlock(&g_crew.m);
  }
  j.parts = parts;
  if (parts > 1) {
    pthread_mutex_lock(&g_crew.m);
    g_crew.job = &j; g_crew.parts = parts;
    __atomic_store_n(&g_crew.pending, parts - 1, __ATOMIC_RELEASE);
    __atomic_add_fetch(&g_crew.generation, 1, __ATOMIC_RELEASE);      /* (the crew looks at these two without the lock while it spins) */
    pthread_cond_broadcast(&g_crew.work);
    pthread_mutex_unlock(&g_crew.m);
  }
  pf_run_part(&j, 0);
  if (parts > 1) {
    for (int spin = 0; spin < 20000 && __atomic_load_n(&g_crew.pending, __ATOMIC_ACQUIRE) != 0; ++spin) __builtin_ia32_pause();
    pthread_mutex_lock(&g_crew.m);
    while (__atomic_load_n(&g_crew.pending, __ATOMIC_ACQUIRE) != 0) pthread_cond_wait(&g_crew.done, &g_crew.m);
    g_crew.job = NULL; g_crew.parts = 0;
    pthread_mutex_unlock(&g_crew.m);
  }
  Py_ssize_t err = -1;
  int norm = 1;
  for (int p = 0; p < parts; ++p) {
    if (j.err[p] >= 0 && (err < 0 || j.err[p] < err)) err = j.err[p];
    if (!j.normalised[p]) norm = 0;
  }
  if (all_normalised) *all_normalised = norm;
  return err;
}

/* set_threads(n): helper threads of the long walks (1 = the calling thread alone); returns the previous setting */
static PyObject* pf_set_threads(PyObject* self, PyObject* args) {
  int nth;
  if (!PyArg_ParseTuple(args, "i", &nth)) return NULL;
  const int prev = pf_crew_size();
  if (nth >= 1) g_walk_threads = nth > PF_MAX_THREADS ? PF_MAX_THREADS : nth;
  return PyLong_FromLong(prev);
}

/* pack_points(seq, dst_addr, capacity_points[, start, count]) -> (n, all_normalised)
 * Copies the 144-byte blobs of seq[start : start + count] (default: all of it) to dst_addr + 144 i, i counted from `start`. */
static PyObject* pf_pack_points(PyObject* self, PyObject* args) {
  PyObject* seq; unsigned long long addr; Py_ssize_t cap, start = 0, count = -1;
  if (!PyArg_ParseTuple(args, "OKn|nn", &seq, &addr, &cap, &start, &count)) return NULL;
  if (!g_point_type) { PyErr_SetString(PyExc_RuntimeError, "_pyface.bind() has not run"); return NULL; }
  PyObject* fast = PySequence_Fast(seq, "pack_points expects a sequence of G1Point");
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  if (count < 0) count = n - start;
  if (start < 0 || start > n || count > n - start) { Py_DECREF(fast); PyErr_SetString(PyExc_ValueError, "range outside the sequence"); return NULL; }
  if (count > cap) { Py_DECREF(fast); PyErr_SetString(PyExc_ValueError, "staging buffer too small"); return NULL; }
  PyObject** items = PySequence_Fast_ITEMS(fast);
  int normalised = 1;
  const Py_ssize_t bad = pf_walk(0, items, start, start + count, n, (uint8_t*)(uintptr_t)addr, &normalised);
  if (bad >= 0) {
    PyObject* o = items[bad];
    if (Py_TYPE(o) != g_point_type) PyErr_Format(PyExc_TypeError, "element %zd is not a G1Point", bad);
    else if (slot_get(o, g_point_off) == Py_None) PyErr_Format(g_unforced, "element %zd is a deferred value", bad);
    else PyErr_Format(PyExc_TypeError, "element %zd holds no 144-byte blob", bad);
    Py_DECREF(fast);
    return NULL;
  }
  Py_DECREF(fast);
  return Py_BuildValue("ni", count, normalised);
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

/* pack_scalars(seq, dst_addr, capacity[, start, count]) -> n
 * Writes int(s) of every Scalar of seq[start : start + count] as 32 little-endian bytes to dst_addr + 32 i (Scalar.to_le_bytes, one call). */
static PyObject* pf_pack_scalars(PyObject* self, PyObject* args) {
  PyObject* seq; unsigned long long addr; Py_ssize_t cap, start = 0, count = -1;
  if (!PyArg_ParseTuple(args, "OKn|nn", &seq, &addr, &cap, &start, &count)) return NULL;
  if (!g_scalar_type) { PyErr_SetString(PyExc_RuntimeError, "_pyface.bind() has not run"); return NULL; }
  PyObject* fast = PySequence_Fast(seq, "pack_scalars expects a sequence of Scalar");
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  if (count < 0) count = n - start;
  if (start < 0 || start > n || count > n - start) { Py_DECREF(fast); PyErr_SetString(PyExc_ValueError, "range outside the sequence"); return NULL; }
  if (count > cap) { Py_DECREF(fast); PyErr_SetString(PyExc_ValueError, "staging buffer too small"); return NULL; }
  PyObject** items = PySequence_Fast_ITEMS(fast);
  uint8_t* dst = (uint8_t*)(uintptr_t)addr;
  Py_ssize_t lo = start;
  const Py_ssize_t hi = start + count;
  while (lo < hi) {
    const Py_ssize_t bad = pf_walk(1, items, lo, hi, n, dst + 32 * (size_t)(lo - start), NULL);
    if (bad < 0) break;
    /* an element the digit reader does not take: wrong type (raise), or an int that is negative / too long / of another digit layout --
     * let the interpreter's own conversion decide (it raises OverflowError where it must), then go on behind it */
    PyObject* o = items[bad];
    PyObject* v;
    if (Py_TYPE(o) == g_scalar_type) v = slot_get(o, g_scalar_off);
    else if (PyLong_CheckExact(o)) v = o;
    else { Py_DECREF(fast); PyErr_Format(PyExc_TypeError, "element %zd is not a Scalar", bad); return NULL; }
    if (!v || !PyLong_Check(v)) { Py_DECREF(fast); PyErr_Format(PyExc_TypeError, "element %zd holds no integer", bad); return NULL; }
    if (pf_long_as_le32(v, dst + 32 * (size_t)(bad - start)) < 0) { Py_DECREF(fast); return NULL; }
    lo = bad + 1;
  }
  Py_DECREF(fast);
  return PyLong_FromSsize_t(count);
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

/* store(points, blobs144 | None, affine96 | None, comp48 | None, clear_terms): the results of one batched evaluation back into the
 * objects -- record i of each buffer becomes a bytes object in the matching slot of points[i] (_blob, _a, _k); clear_terms: _t = None.
 * (The Python loop this replaces cost ~0.5 us per object and slot: 0.3 ms for the 585 points of one verification.) */
static void slot_replace(PyObject* o, Py_ssize_t off, PyObject* v /* reference stolen */) {
  PyObject** p = (PyObject**)((char*)o + off);
  PyObject* old = *p;
  *p = v;
  Py_XDECREF(old);
}
static PyObject* pf_store(PyObject* self, PyObject* args) {
  PyObject *seq, *src[3];
  int clear_t = 0;
  if (!PyArg_ParseTuple(args, "OOOOp", &seq, &src[0], &src[1], &src[2], &clear_t)) return NULL;
  if (!g_point_type || g_t_off < 0) { PyErr_SetString(PyExc_RuntimeError, "_pyface.bind() has not run"); return NULL; }
  const Py_ssize_t rec[3] = {144, 96, 48};
  const Py_ssize_t off[3] = {g_point_off, g_cache_off[0], g_cache_off[1]};
  PyObject* fast = PySequence_Fast(seq, "store: points must be a sequence");
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  PyObject** items = PySequence_Fast_ITEMS(fast);
  Py_buffer view[3];
  int have[3] = {0, 0, 0}, ok = 1;
  for (int k = 0; k < 3 && ok; ++k) {
    if (src[k] == Py_None) continue;
    if (off[k] < 0 || PyObject_GetBuffer(src[k], &view[k], PyBUF_SIMPLE) < 0) { if (off[k] < 0) PyErr_SetString(PyExc_RuntimeError, "slot not bound"); ok = 0; break; }
    have[k] = 1;
    if (view[k].len < n * rec[k]) { PyErr_SetString(PyExc_ValueError, "store: buffer shorter than the point list"); ok = 0; }
  }
  for (Py_ssize_t i = 0; i < n && ok; ++i) {
    PyObject* o = items[i];
    if (Py_TYPE(o) != g_point_type) { PyErr_SetString(PyExc_TypeError, "store: not a G1Point"); ok = 0; break; }
    for (int k = 0; k < 3; ++k) {
      if (!have[k]) continue;
      PyObject* v = PyBytes_FromStringAndSize((const char*)view[k].buf + i * rec[k], rec[k]);
      if (!v) { ok = 0; break; }
      slot_replace(o, off[k], v);
    }
    if (ok && clear_t) { Py_INCREF(Py_None); slot_replace(o, g_t_off, Py_None); }
  }
  for (int k = 0; k < 3; ++k) if (have[k]) PyBuffer_Release(&view[k]);
  Py_DECREF(fast);
  if (!ok) return NULL;
  Py_RETURN_NONE;
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

/* ---- deferred values: the coefficient bookkeeping of py_arkworks_bls12381.py in C (integers stay Python ints: one multiply + one
 * remainder per coefficient through the number protocol, without a bytecode per element) */

/* scale(coefs, v, R) -> [c * v % R for c in coefs] */
static PyObject* pf_scale(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 3 || !PyList_CheckExact(args[0])) { PyErr_SetString(PyExc_TypeError, "scale(list, v, R)"); return NULL; }
  PyObject *src = args[0], *v = args[1], *R = args[2];
  const Py_ssize_t n = PyList_GET_SIZE(src);
  PyObject* out = PyList_New(n);
  if (!out) return NULL;
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject* m = PyNumber_Multiply(PyList_GET_ITEM(src, i), v);
    PyObject* r = m ? PyNumber_Remainder(m, R) : NULL;
    Py_XDECREF(m);
    if (!r) { Py_DECREF(out); return NULL; }
    PyList_SET_ITEM(out, i, r);
  }
  return out;
}

/* msm_terms(bases, scalars, n, R) -> (coefs, leaves, all_leaves_in_g1): sum_i scalars[i] * bases[i] as one coefficient list over
 * leaves; a deferred base (its `_t` = (coefs, leaves, from_msm); the caller made sure its leaves are in G1) contributes its terms with
 * the scalar folded in mod R.  TypeError for anything that is not a G1Point / Scalar (or an int in [0, R)). */
static PyObject* pf_msm_terms(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 4) { PyErr_SetString(PyExc_TypeError, "msm_terms(bases, scalars, n, R)"); return NULL; }
  if (!g_point_type || !g_scalar_type || g_t_off < 0 || g_sg_off < 0) { PyErr_SetString(PyExc_RuntimeError, "_pyface.bind() has not run"); return NULL; }
  PyObject* fb = PySequence_Fast(args[0], "msm_terms expects sequences");
  if (!fb) return NULL;
  PyObject* fs = PySequence_Fast(args[1], "msm_terms expects sequences");
  if (!fs) { Py_DECREF(fb); return NULL; }
  const Py_ssize_t n = PyLong_AsSsize_t(args[2]);
  PyObject* R = args[3];
  PyObject *coefs = NULL, *leaves = NULL, *zero = NULL, *unsure = NULL;
  int all_g1 = 1;
  if (n < 0 || n > PySequence_Fast_GET_SIZE(fb) || n > PySequence_Fast_GET_SIZE(fs)) { PyErr_SetString(PyExc_ValueError, "msm_terms: n outside the sequences"); goto fail; }
  coefs = PyList_New(0); leaves = PyList_New(0); zero = PyLong_FromLong(0);
  if (!coefs || !leaves || !zero) goto fail;
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject* b = PySequence_Fast_GET_ITEM(fb, i);
    PyObject* sc = PySequence_Fast_GET_ITEM(fs, i);
    PyObject* v;
    if (Py_TYPE(b) != g_point_type) { PyErr_Format(PyExc_TypeError, "compute_MSM: bases must be G1Point, not %s", Py_TYPE(b)->tp_name); goto fail; }
    if (Py_TYPE(sc) == g_scalar_type) v = slot_get(sc, g_scalar_off);
    else if (PyLong_CheckExact(sc)) {
      v = sc;
      const int neg = PyObject_RichCompareBool(v, zero, Py_LT), big = PyObject_RichCompareBool(v, R, Py_GE);
      if (neg < 0 || big < 0) goto fail;
      if (neg || big) { PyErr_SetString(PyExc_TypeError, "compute_MSM: scalars must be Scalar"); goto fail; }
    } else { PyErr_Format(PyExc_TypeError, "compute_MSM: scalars must be Scalar, not %s", Py_TYPE(sc)->tp_name); goto fail; }
    if (!v || !PyLong_Check(v)) { PyErr_SetString(PyExc_TypeError, "compute_MSM: a Scalar holds no integer"); goto fail; }
    PyObject* t = slot_get(b, g_t_off);
    if (!t || t == Py_None) {
      if (PyList_Append(coefs, v) < 0 || PyList_Append(leaves, b) < 0) goto fail;
      if (slot_get(b, g_sg_off) != Py_True) all_g1 = 0;
    } else {
      if (!PyTuple_CheckExact(t) || PyTuple_GET_SIZE(t) != 3) { PyErr_SetString(PyExc_TypeError, "malformed deferred value"); goto fail; }
      if (slot_get(b, g_sg_off) != Py_True) {
        /* a deferred base over leaves not known to be in G1: the caller has them tested (or evaluates the base) and calls again */
        if (!unsure) { unsure = PyList_New(0); if (!unsure) goto fail; }
        if (PyList_Append(unsure, b) < 0) goto fail;
        continue;
      }
      if (unsure) continue;
      PyObject *cs = PyTuple_GET_ITEM(t, 0), *ls = PyTuple_GET_ITEM(t, 1);
      if (!PyList_CheckExact(cs) || !PyList_CheckExact(ls) || PyList_GET_SIZE(cs) != PyList_GET_SIZE(ls)) { PyErr_SetString(PyExc_TypeError, "malformed deferred value"); goto fail; }
      const Py_ssize_t k = PyList_GET_SIZE(cs);
      for (Py_ssize_t j = 0; j < k; ++j) {
        PyObject* m = PyNumber_Multiply(PyList_GET_ITEM(cs, j), v);
        PyObject* r = m ? PyNumber_Remainder(m, R) : NULL;
        Py_XDECREF(m);
        if (!r) goto fail;
        const int rc = PyList_Append(coefs, r);
        Py_DECREF(r);
        if (rc < 0 || PyList_Append(leaves, PyList_GET_ITEM(ls, j)) < 0) goto fail;
      }
    }
  }
  Py_DECREF(fb); Py_DECREF(fs); Py_DECREF(zero);
  if (unsure) { Py_DECREF(coefs); Py_DECREF(leaves); return Py_BuildValue("OOON", Py_None, Py_None, Py_None, unsure); }
  return Py_BuildValue("NNOO", coefs, leaves, all_g1 ? Py_True : Py_False, Py_None);
fail:
  Py_DECREF(fb); Py_DECREF(fs); Py_XDECREF(coefs); Py_XDECREF(leaves); Py_XDECREF(zero); Py_XDECREF(unsure);
  return NULL;
}

/* assemble(nodes, R) -> (leaf_list, offsets, term_base, scalars32, from_msm): the arrays cg1_lincomb_batch takes for a flush of the
 * deferred values `nodes` (each `_t` = (coefs, leaves, from_msm), `_sg` True = every leaf in G1).  Over G1 leaves a coefficient goes
 * in as c mod R; otherwise it stays the integer it is (|c| < R), the sign moving to the base (bit 31 of term_base).  Zero coefficients
 * are dropped; equal leaves share one base index. */
static PyObject* pf_assemble(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 2) { PyErr_SetString(PyExc_TypeError, "assemble(nodes, R)"); return NULL; }
  if (!g_point_type || g_t_off < 0 || g_sg_off < 0) { PyErr_SetString(PyExc_RuntimeError, "_pyface.bind() has not run"); return NULL; }
  PyObject* fn = PySequence_Fast(args[0], "assemble expects a sequence of deferred values");
  if (!fn) return NULL;
  PyObject* R = args[1];
  const Py_ssize_t n_out = PySequence_Fast_GET_SIZE(fn);
  Py_ssize_t total = 0;
  for (Py_ssize_t j = 0; j < n_out; ++j) {
    PyObject* nd = PySequence_Fast_GET_ITEM(fn, j);
    PyObject* t = Py_TYPE(nd) == g_point_type ? slot_get(nd, g_t_off) : NULL;
    if (!t || !PyTuple_CheckExact(t) || PyTuple_GET_SIZE(t) != 3 || !PyList_CheckExact(PyTuple_GET_ITEM(t, 0)) || !PyList_CheckExact(PyTuple_GET_ITEM(t, 1)) ||
        PyList_GET_SIZE(PyTuple_GET_ITEM(t, 0)) != PyList_GET_SIZE(PyTuple_GET_ITEM(t, 1))) {
      Py_DECREF(fn); PyErr_Format(PyExc_TypeError, "element %zd is not a deferred value", j); return NULL;
    }
    total += PyList_GET_SIZE(PyTuple_GET_ITEM(t, 0));
  }
  size_t cap = 16;
  while (cap < (size_t)total * 2 + 2) cap <<= 1;
  PyObject** keys = (PyObject**)calloc(cap, sizeof(PyObject*));
  uint32_t* vals = (uint32_t*)malloc(cap * sizeof(uint32_t));
  PyObject* offs = PyBytes_FromStringAndSize(NULL, (n_out + 1) * 4);
  PyObject* tb = PyBytes_FromStringAndSize(NULL, (total ? total : 1) * 4);
  PyObject* sc = PyBytes_FromStringAndSize(NULL, (total ? total : 1) * 32);
  PyObject* leaf_list = PyList_New(0);
  PyObject* zero = PyLong_FromLong(0);
  int from_msm = 0;
  Py_ssize_t T = 0;
  if (!keys || !vals || !offs || !tb || !sc || !leaf_list || !zero) { PyErr_NoMemory(); goto fail; }
  {
    uint32_t* po = (uint32_t*)PyBytes_AS_STRING(offs);
    uint32_t* pt = (uint32_t*)PyBytes_AS_STRING(tb);
    uint8_t* ps = (uint8_t*)PyBytes_AS_STRING(sc);
    po[0] = 0;
    for (Py_ssize_t j = 0; j < n_out; ++j) {
      PyObject* nd = PySequence_Fast_GET_ITEM(fn, j);
      PyObject* t = slot_get(nd, g_t_off);
      PyObject *cs = PyTuple_GET_ITEM(t, 0), *ls = PyTuple_GET_ITEM(t, 1);
      if (PyObject_IsTrue(PyTuple_GET_ITEM(t, 2))) from_msm = 1;
      const int in_g1 = slot_get(nd, g_sg_off) == Py_True;
      const Py_ssize_t k = PyList_GET_SIZE(cs);
      for (Py_ssize_t i = 0; i < k; ++i) {
        PyObject* c = PyList_GET_ITEM(cs, i);
        PyObject* leaf = PyList_GET_ITEM(ls, i);
        PyObject* mag = NULL;
        uint32_t negbit = 0;
        if (in_g1) {
          mag = PyNumber_Remainder(c, R);                      /* non-negative for a positive R */
        } else {
          const int neg = PyObject_RichCompareBool(c, zero, Py_LT);
          if (neg < 0) goto fail;
          if (neg) { mag = PyNumber_Negative(c); negbit = 0x80000000u; } else { mag = c; Py_INCREF(mag); }
        }
        if (!mag) goto fail;
        const int is_zero = PyObject_RichCompareBool(mag, zero, Py_EQ);
        if (is_zero < 0) { Py_DECREF(mag); goto fail; }
        if (is_zero) { Py_DECREF(mag); continue; }
        if (!PyLong_Check(mag) || (!long_to_le32(mag, ps + 32 * T) && pf_long_as_le32(mag, ps + 32 * T) < 0)) { Py_DECREF(mag); if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "coefficient is not an int"); goto fail; }
        Py_DECREF(mag);
        size_t h = ((size_t)(uintptr_t)leaf >> 4) * 0x9E3779B97F4A7C15ull;
        h = (h >> 17) & (cap - 1);
        while (keys[h] && keys[h] != leaf) h = (h + 1) & (cap - 1);
        if (!keys[h]) {
          keys[h] = leaf;
          vals[h] = (uint32_t)PyList_GET_SIZE(leaf_list);
          if (PyList_Append(leaf_list, leaf) < 0) goto fail;
        }
        pt[T] = vals[h] | negbit;
        ++T;
      }
      po[j + 1] = (uint32_t)T;
    }
  }
  free(keys); free(vals); Py_DECREF(fn); Py_DECREF(zero);
  return Py_BuildValue("NNNNnO", leaf_list, offs, tb, sc, T, from_msm ? Py_True : Py_False);
fail:
  free(keys); free(vals); Py_DECREF(fn); Py_XDECREF(offs); Py_XDECREF(tb); Py_XDECREF(sc); Py_XDECREF(leaf_list); Py_XDECREF(zero);
  return NULL;
}

static PyMethodDef methods[] = {
    {"scale", (PyCFunction)(void (*)(void))pf_scale, METH_FASTCALL, "scale(coefs, v, R) -> [c * v % R]"},
    {"msm_terms", (PyCFunction)(void (*)(void))pf_msm_terms, METH_FASTCALL, "msm_terms(bases, scalars, n, R) -> (coefs, leaves, all_in_g1, None) | (None, None, None, [deferred bases whose leaves must be tested first])"},
    {"assemble", (PyCFunction)(void (*)(void))pf_assemble, METH_FASTCALL, "assemble(nodes, R) -> (leaves, offsets, term_base, scalars32, T, from_msm)"},
    {"store", pf_store, METH_VARARGS, "store(points, blobs144 | None, affine96 | None, comp48 | None, clear_terms)"},
    {"mk", (PyCFunction)(void (*)(void))pf_mk, METH_FASTCALL, "mk(blob, a, k, t, sg, seq) -> G1Point"},
    {"decode_lazy", pf_decode_lazy, METH_O, "decode_lazy(data48) -> G1Point (validated, y deferred)"},
    {"set_native", pf_set_native, METH_VARARGS, "set_native(address of cg1_validate_compressed)"},
    {"set_threads", pf_set_threads, METH_VARARGS, "set_threads(n) -> previous: threads of the long walks"},
    {"ident", pf_ident, METH_VARARGS, "ident(seq) -> (n, fingerprint of the element identities)"},
    {"same_items", pf_same_items, METH_VARARGS, "same_items(a, b) -> bool"},
    {"bind", pf_bind, METH_VARARGS, "bind(G1Point, Scalar)"},
    {"pack_points", pf_pack_points, METH_VARARGS, "pack_points(seq, dst_addr, capacity[, start, count]) -> (n, all_normalised)"},
    {"pack_affine", pf_pack_affine, METH_VARARGS, "pack_affine(seq, dst_addr, capacity) -> n"},
    {"pack_scalars", pf_pack_scalars, METH_VARARGS, "pack_scalars(seq, dst_addr, capacity[, start, count]) -> n"},
    {"points_from_blobs", pf_points_from_blobs, METH_VARARGS, "points_from_blobs(data, n) -> list of G1Point"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pyface", "marshalling helper of the curdleproofs_pie_amd Python face", -1, methods};

PyMODINIT_FUNC PyInit__pyface(void) {
  PyObject* m = PyModule_Create(&moddef);
  if (!m) return NULL;
  pthread_atfork(NULL, NULL, pf_crew_after_fork);
  g_unforced = PyErr_NewException("_pyface.Unforced", PyExc_LookupError, NULL);
  if (!g_unforced) { Py_DECREF(m); return NULL; }
  Py_INCREF(g_unforced);
  if (PyModule_AddObject(m, "Unforced", g_unforced) < 0) { Py_DECREF(g_unforced); Py_DECREF(m); return NULL; }
  return m;
}
