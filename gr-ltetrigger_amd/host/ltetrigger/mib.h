// gr::ltetrigger::mib -- PBCH/MIB block (reference: include/ltetrigger/mib.h:36-49).
#pragma once
#include <memory>

#include "gr_api.h"
#include "api.h"

namespace gr {
namespace ltetrigger {

class LTETRIGGER_API mib : virtual public gr::block
{
public:
  typedef std::shared_ptr<mib> sptr;
  static sptr make(bool exit_on_success = false);

protected:
  mib() : gr::block("mib", nullptr, nullptr) {}
};

} // namespace ltetrigger
} // namespace gr
