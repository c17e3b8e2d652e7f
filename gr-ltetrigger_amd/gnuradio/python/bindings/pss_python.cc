// gr::ltetrigger::pss for Python (include/ltetrigger/pss.h:36-87 of the reference: make + six methods).
// NOT COMPILED IN THIS REPOSITORY (no GNU Radio).
#include <pybind11/pybind11.h>

#include <ltetrigger/pss.h>

namespace py = pybind11;

void bind_pss(py::module &m)
{
  using pss = ::gr::ltetrigger::pss;
  py::class_<pss, gr::block, gr::basic_block, std::shared_ptr<pss>>(m, "pss",
      "Primary synchronisation: PSS search on an MI355X, peak-to-side-lobe score, tracking, CFO correction; emits "
      "half-frames aligned to subframe 0 / 5 (input and output at 1.92 Msps)")
      .def(py::init(&pss::make), py::arg("N_id_2"), py::arg("psr_threshold"), py::arg("track_after") = 16,
           py::arg("track_every") = 8)
      .def("max_psr", &pss::max_psr)
      .def("mean_psr", &pss::mean_psr)
      .def("mean_cfo", &pss::mean_cfo)
      .def("set_psr_threshold", &pss::set_psr_threshold, py::arg("threshold"))
      .def("psr_threshold", &pss::psr_threshold)
      .def("tracking_score", &pss::tracking_score);
}
