#pragma once
#include <list>
#include <mutex>

#include "ltetrigger/cellstore.h"

namespace gr {
namespace ltetrigger {

class cellstore_impl : public cellstore
{
public:
  cellstore_impl();
  bool tracking() override;
  std::vector<pmt::pmt_t> cells() override;
  pmt::pmt_t latest_cell() override;

private:
  void track_cell(pmt::pmt_t msg);
  void drop_cell(pmt::pmt_t msg);
  std::mutex d_mutex;
  std::list<pmt::pmt_t> d_cells;
};

} // namespace ltetrigger
} // namespace gr
