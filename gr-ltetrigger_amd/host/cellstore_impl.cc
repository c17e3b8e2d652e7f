// gr::ltetrigger::cellstore: a mutex-guarded list fed by the "track" / "drop" message ports
// (reference behaviour: lib/cellstore_impl.cc:60-105 -- drop removes by object identity).
#include "cellstore_impl.h"

#include <stdexcept>

namespace gr {
namespace ltetrigger {

cellstore::sptr cellstore::make() { return gnuradio::get_initial_sptr(new cellstore_impl); }

cellstore_impl::cellstore_impl() : gr::basic_block("cellstore")
{
  message_port_register_in(pmt::intern("track"));
  set_msg_handler(pmt::intern("track"), [this](pmt::pmt_t m) { track_cell(m); });
  message_port_register_in(pmt::intern("drop"));
  set_msg_handler(pmt::intern("drop"), [this](pmt::pmt_t m) { drop_cell(m); });
}

bool cellstore_impl::tracking()
{
  std::lock_guard<std::mutex> lock(d_mutex);
  return !d_cells.empty();
}

std::vector<pmt::pmt_t> cellstore_impl::cells()
{
  std::lock_guard<std::mutex> lock(d_mutex);
  return std::vector<pmt::pmt_t>(d_cells.begin(), d_cells.end());
}

pmt::pmt_t cellstore_impl::latest_cell()
{
  std::lock_guard<std::mutex> lock(d_mutex);
  return d_cells.empty() ? PMT_NIL : d_cells.back();
}

void cellstore_impl::track_cell(pmt::pmt_t msg)
{
  const pmt::pmt_t bad = pmt::from_long(1000);
  if (!pmt::is_dict(msg) || pmt::to_long(pmt::dict_ref(msg, pmt::intern("cell_id"), bad)) == 1000)
    throw std::runtime_error("Error tracking cell: bad message format");
  std::lock_guard<std::mutex> lock(d_mutex);
  d_cells.push_back(msg);
}

void cellstore_impl::drop_cell(pmt::pmt_t msg)
{
  std::lock_guard<std::mutex> lock(d_mutex);
  d_cells.remove_if([&](const pmt::pmt_t &c) { return pmt::eq(c, msg); });
}

} // namespace ltetrigger
} // namespace gr
