// gr::ltetrigger::sss for Python (include/ltetrigger/sss.h:36-51 of the reference).  NOT COMPILED IN THIS REPOSITORY (no GNU Radio).
#include <pybind11/pybind11.h>

#include <ltetrigger/sss.h>

namespace py = pybind11;

void bind_sss(py::module &m)
{
  using sss = ::gr::ltetrigger::sss;
  py::class_<sss, gr::sync_block, gr::block, gr::basic_block, std::shared_ptr<sss>>(m, "sss",
      "Secondary synchronisation on the half-frames the pss block emits: CP length, SSS m0/m1, N_id_1; tags cell_id and "
      "cp_type on the first item of every tracked half-frame")
      .def(py::init(&sss::make), py::arg("N_id_2"));
}
