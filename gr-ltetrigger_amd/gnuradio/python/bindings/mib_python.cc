// gr::ltetrigger::mib for Python (include/ltetrigger/mib.h:36-49 of the reference).  NOT COMPILED IN THIS REPOSITORY (no GNU Radio).
#include <pybind11/pybind11.h>

#include <ltetrigger/mib.h>

namespace py = pybind11;

void bind_mib(py::module &m)
{
  using mib = ::gr::ltetrigger::mib;
  py::class_<mib, gr::block, gr::basic_block, std::shared_ptr<mib>>(m, "mib",
      "PBCH / MIB decode of tagged half-frames; message ports \"track\" and \"drop\" carry one PMT dict per cell")
      .def(py::init(&mib::make), py::arg("exit_on_success") = false);
}
