// gr::ltetrigger::sss -- secondary synchronisation block (reference: include/ltetrigger/sss.h:36-51).
#pragma once
#include <memory>

#include "gr_api.h"
#include "api.h"

namespace gr {
namespace ltetrigger {

class LTETRIGGER_API sss : virtual public gr::sync_block
{
public:
  typedef std::shared_ptr<sss> sptr;
  static sptr make(int N_id_2);

protected:
  sss() : gr::sync_block("sss", nullptr, nullptr) {}
};

} // namespace ltetrigger
} // namespace gr
