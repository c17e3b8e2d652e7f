// gr::ltetrigger::cellstore -- holds the cells currently tracked
// (reference: include/ltetrigger/cellstore.h:43-65).
#pragma once
#include <memory>
#include <vector>

#include "gr_api.h"
#include "api.h"

namespace gr {
namespace ltetrigger {

class LTETRIGGER_API cellstore : virtual public gr::basic_block
{
public:
  typedef std::shared_ptr<cellstore> sptr;
  static sptr make();

  virtual bool tracking() = 0;                   // any cell tracked right now?
  virtual std::vector<pmt::pmt_t> cells() = 0;   // all tracked cells, oldest first
  virtual pmt::pmt_t latest_cell() = 0;          // most recent one or PMT_NIL

protected:
  cellstore() : gr::basic_block("cellstore") {}
};

} // namespace ltetrigger
} // namespace gr
