// gr::ltetrigger::pss -- primary synchronisation block, public interface.
// Same class name, namespace, make() signature and getters as the reference's
// include/ltetrigger/pss.h:36-87; the implementation calls the gfx950 library instead of srsLTE.
#pragma once
#include <memory>

#include "gr_api.h"
#include "api.h"

namespace gr {
namespace ltetrigger {

class LTETRIGGER_API pss : virtual public gr::block
{
public:
  typedef std::shared_ptr<pss> sptr;

  // N_id_2: PSS root index this block searches for.
  // psr_threshold: a half-frame whose peak-to-side-lobe ratio exceeds it adds one to the tracking
  //   score; one that does not resets the score and, if the block was tracking, signals
  //   "tracking_lost" downstream.
  // track_after: consecutive hits needed to enter tracking.  track_every: while tracking, search
  //   only once per this many half-frames.
  static sptr make(int N_id_2, float psr_threshold, int track_after = 16, int track_every = 8);

  virtual float max_psr() const = 0;        // largest PSR seen
  virtual float mean_psr() const = 0;       // mean PSR over the last <= 200 searches
  virtual float mean_cfo() const = 0;       // mean fractional CFO (15 kHz units) over <= 200 estimates
  virtual void set_psr_threshold(float threshold) = 0;
  virtual float psr_threshold() const = 0;
  virtual float tracking_score() const = 0;

protected:
  pss() : gr::block("pss", nullptr, nullptr) {}
};

} // namespace ltetrigger
} // namespace gr
