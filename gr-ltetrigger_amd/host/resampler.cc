#include "resampler.h"

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>

namespace gr {
namespace filter {

rational_resampler_ccc::sptr rational_resampler_ccc::make(unsigned interpolation, unsigned decimation, int device)
{
  if (interpolation != 1) throw std::invalid_argument("rational_resampler_ccc: only interpolation = 1 is needed by this path");
  return sptr(new rational_resampler_ccc(decimation, device));
}

rational_resampler_ccc::rational_resampler_ccc(unsigned decimation, int device)
    : gr::block("rational_resampler_ccc", gr::io_signature::make(1, 1, sizeof(lt_cf32)),
                gr::io_signature::make(1, 1, sizeof(lt_cf32))),
      d_D((int)decimation)
{
  if (lt_decim_create(&d_decim, device, d_D))
    throw std::runtime_error(std::string("lt_decim_create: ") + lt_last_error());
  const int ntaps = lt_decim_ntaps(d_decim);
  d_hist = ((ntaps - 1 + d_D - 1) / d_D) * d_D;
  set_history((unsigned)d_hist + 1);
}

rational_resampler_ccc::~rational_resampler_ccc() { lt_decim_free(d_decim); }

void rational_resampler_ccc::forecast(int noutput_items, gr_vector_int &required)
{
  for (auto &r : required) r = noutput_items * d_D + (int)history() - 1;
}

int rational_resampler_ccc::general_work(int noutput_items, gr_vector_int &ninput_items,
                                         gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
{
  // input_items[0] starts d_hist items before the first unconsumed one; with d_hist a multiple of
  // D the output phase is unchanged and outputs k' >= d_hist/D have their full filter history.
  const lt_cf32 *in = static_cast<const lt_cf32 *>(input_items[0]);
  lt_cf32 *out = static_cast<lt_cf32 *>(output_items[0]);
  const int fresh = ninput_items[0] - d_hist;
  const int nout = std::min(noutput_items, fresh / d_D);
  if (nout <= 0) return 0;
  const int64_t n_in = (int64_t)d_hist + (int64_t)nout * d_D;
  d_tmp.resize((size_t)(n_in / d_D));
  if (lt_decim_run(d_decim, in, d_tmp.data(), 1, n_in, 0, nullptr))
    throw std::runtime_error(std::string("lt_decim_run: ") + lt_last_error());
  std::memcpy(out, d_tmp.data() + d_hist / d_D, sizeof(lt_cf32) * (size_t)nout);
  consume_each(nout * d_D);
  return nout;
}

} // namespace filter
} // namespace gr
