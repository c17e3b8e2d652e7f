// pybind11 module of the ltetrigger blocks for GNU Radio >= 3.9 (the layout `gr_modtool bind` generates).
// Binds what the reference binds with SWIG (swig/ltetrigger_swig.i:19-29): pss, sss, mib, cellstore.
// NOT COMPILED IN THIS REPOSITORY: no GNU Radio in either container (see ../../README.md).
#include <pybind11/pybind11.h>

#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

namespace py = pybind11;

void bind_pss(py::module &m);
void bind_sss(py::module &m);
void bind_mib(py::module &m);
void bind_cellstore(py::module &m);

// numpy's C API must be initialised once per module (GNU Radio's own bindings do the same)
void *init_numpy()
{
  import_array();
  return NULL;
}

PYBIND11_MODULE(ltetrigger_python, m)
{
  init_numpy();
  // the block base classes and PMT come from GNU Radio's own modules
  py::module::import("gnuradio.gr");
  py::module::import("pmt");

  bind_pss(m);
  bind_sss(m);
  bind_mib(m);
  bind_cellstore(m);
}
