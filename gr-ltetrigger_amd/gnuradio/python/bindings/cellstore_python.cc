// gr::ltetrigger::cellstore for Python (include/ltetrigger/cellstore.h:43-65 of the reference).  NOT COMPILED IN THIS REPOSITORY (no GNU Radio).
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <ltetrigger/cellstore.h>

namespace py = pybind11;

void bind_cellstore(py::module &m)
{
  using cellstore = ::gr::ltetrigger::cellstore;
  py::class_<cellstore, gr::basic_block, std::shared_ptr<cellstore>>(m, "cellstore",
      "Keeps the cells announced on \"track\" until they are dropped")
      .def(py::init(&cellstore::make))
      .def("tracking", &cellstore::tracking)
      .def("cells", &cellstore::cells)           // std::vector<pmt::pmt_t>: the reference's %template() line in the SWIG file
      .def("latest_cell", &cellstore::latest_cell);
}
