/* pyhandoff.c - the hand-off of the filtration to Python objects (host C, CPython API, no GPU).
 *
 * flood_complex returns dict[tuple[int, ...], float] (reference flooder/core.py:258-263, 285-288: the dict is
 * filled from zip(faces.tolist(), values.tolist()) per batch and rebuilt from stree.get_simplices() at the end).
 * With the sweep at ~1 ms the 26 k tuples of a 1000-landmark complex were 3.6 ms of the call when built through
 * .tolist() + zip; here they are made in one pass over the integer table: the int object of a vertex id is created
 * once and shared by every tuple that holds it.
 *
 * Loaded with ctypes.PyDLL (the GIL is held for the call).
 *   flooder_dict_update(dict, rows int64 (n, k) row-major, n, k, vals float64 (n,), cache list)  ->  0 / -1
 *   cache: a Python list whose entry v is the int object v (grown here as needed; pass the same list for all tables
 *   of one complex).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

int flooder_dict_update(PyObject* dict, const int64_t* rows, int64_t n, int k, const double* vals, PyObject* cache) {
  if (!dict || !PyDict_Check(dict) || !cache || !PyList_Check(cache) || n < 0 || k < 1 || (n > 0 && (!rows || !vals))) {
    PyErr_SetString(PyExc_ValueError, "flooder_dict_update: bad argument");
    return -1;
  }
  for (int64_t i = 0; i < n; ++i) {
    PyObject* key = PyTuple_New(k);
    if (!key) return -1;
    for (int j = 0; j < k; ++j) {
      const int64_t v = rows[i * k + j];
      PyObject* o;
      if (v >= 0 && v < (int64_t)1 << 24) {
        Py_ssize_t have = PyList_GET_SIZE(cache);
        while (have <= (Py_ssize_t)v) {  /* grow the cache up to v */
          PyObject* nv = PyLong_FromSsize_t(have);
          if (!nv || PyList_Append(cache, nv) < 0) { Py_XDECREF(nv); Py_DECREF(key); return -1; }
          Py_DECREF(nv);
          ++have;
        }
        o = PyList_GET_ITEM(cache, (Py_ssize_t)v);
        Py_INCREF(o);
      } else {
        o = PyLong_FromLongLong((long long)v);
        if (!o) { Py_DECREF(key); return -1; }
      }
      PyTuple_SET_ITEM(key, j, o);
    }
    PyObject* val = PyFloat_FromDouble(vals[i]);
    if (!val) { Py_DECREF(key); return -1; }
    const int rc = PyDict_SetItem(dict, key, val);
    Py_DECREF(key);
    Py_DECREF(val);
    if (rc < 0) return -1;
  }
  return 0;
}
