#include "runtime.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <thread>

namespace gr {

// ---------------------------------------------------------------- messages
void basic_block::message_port_pub(pmt::pmt_t port, pmt::pmt_t msg)
{
  auto it = d_out_ports.find(port.get());
  if (it == d_out_ports.end()) throw std::runtime_error("message_port_pub: port not registered on " + d_name);
  for (auto &sub : it->second) sub.first->post(sub.second, msg);
}

void basic_block::post(pmt::pmt_t port, pmt::pmt_t msg)
{
  auto it = d_in_ports.find(port.get());
  if (it == d_in_ports.end() || !it->second) return;
  std::lock_guard<std::mutex> lock(d_msg_mu);
  it->second(msg);
}

// ---------------------------------------------------------------- block
namespace {
constexpr int kDefaultChunk = 8192;
constexpr size_t kHighWater = 1 << 16;
} // namespace

block::block(const std::string &name, io_signature::sptr in, io_signature::sptr out)
    : basic_block(name), d_in_sig(in), d_out_sig(out)
{
}

void block::set_history(unsigned h)
{
  if (h < 1) h = 1;
  // zero history at stream start: pre-fill with h-1 zeros in front of item 0
  if (d_nread != 0 || (!d_in.empty() && d_in_abs0 + (int64_t)d_in.size() > 0))
    throw std::runtime_error("set_history after start is not supported");
  d_history = h;
  d_in.assign(h - 1, gr_complex(0, 0));
  d_in_abs0 = -(int64_t)(h - 1);
}

void block::forecast(int noutput_items, gr_vector_int &required)
{
  for (auto &r : required) r = noutput_items + (int)history() - 1;
}

void block::add_item_tag(unsigned, uint64_t abs_offset, const pmt::pmt_t &key, const pmt::pmt_t &value,
                         const pmt::pmt_t &srcid)
{
  tag_t t;
  t.offset = abs_offset;
  t.key = key;
  t.value = value;
  t.srcid = srcid;
  d_new_tags.push_back(t);
}

void block::get_tags_in_range(std::vector<tag_t> &v, unsigned, uint64_t abs_start, uint64_t abs_end)
{
  v.clear();
  for (const tag_t &t : d_in_tags)
    if (t.offset >= abs_start && t.offset < abs_end) v.push_back(t);
}

void block::get_tags_in_window(std::vector<tag_t> &v, unsigned which, uint64_t rel_start, uint64_t rel_end,
                               const pmt::pmt_t &key)
{
  std::vector<tag_t> all;
  get_tags_in_range(all, which, (uint64_t)d_nread + rel_start, (uint64_t)d_nread + rel_end);
  v.clear();
  for (const tag_t &t : all)
    if (pmt::eq(t.key, key)) v.push_back(t);
}

// (called on the upstream's thread)
void block::accept(const gr_complex *items, size_t n, const std::vector<tag_t> &tags)
{
  std::lock_guard<std::mutex> lock(d_mu);
  d_inbox.insert(d_inbox.end(), items, items + n);
  d_inbox_tags.insert(d_inbox_tags.end(), tags.begin(), tags.end());
  d_backlog.fetch_add(n, std::memory_order_relaxed);
}

// (on the block's own thread: whatever the upstream has handed over since the last step)
void block::take_inbox()
{
  // "the upstream is done" is read BEFORE the inbox is emptied: its last items were handed over before it said so
  if (d_upstream && d_upstream->d_done.load(std::memory_order_acquire)) d_upstream_done = true;
  std::lock_guard<std::mutex> lock(d_mu);
  if (!d_inbox.empty()) {
    d_in.insert(d_in.end(), d_inbox.begin(), d_inbox.end());
    d_inbox.clear();
  }
  if (!d_inbox_tags.empty()) {
    d_in_tags.insert(d_in_tags.end(), d_inbox_tags.begin(), d_inbox_tags.end());
    d_inbox_tags.clear();
  }
}

bool block::step()
{
  if (d_done) return false;
  if (!d_downstream.empty()) {
    bool all_done = true;
    for (block *d : d_downstream) all_done = all_done && d->d_done;
    if (all_done) {
      d_done.store(true, std::memory_order_release);
      return true;
    }
    for (block *d : d_downstream) // back-pressure
      if (!d->d_done && d->backlog() > kHighWater + (size_t)d->d_output_multiple + d->d_history) return false;
  }
  const bool source = is_source();
  int noutput = d_output_multiple > 1 ? d_output_multiple : kDefaultChunk;
  int64_t avail = 0;
  d_draining = false;
  if (!source) take_inbox();
  if (!source) {
    avail = d_in_abs0 + (int64_t)d_in.size() - d_nread;
    const int64_t have = avail + (int64_t)history() - 1;
    gr_vector_int required(1, 0);
    bool fits = false;
    if (d_output_multiple > 1) {
      // one output multiple per call keeps the call sequence independent of buffer fill levels
      forecast(noutput, required);
      fits = have >= required[0];
    } else {
      // like GNU Radio's executor: shrink the request until forecast() is satisfied
      for (noutput = (int)std::min<int64_t>(kDefaultChunk, std::max<int64_t>(avail, 1)); noutput >= 1; noutput /= 2) {
        forecast(noutput, required);
        if (have >= required[0]) {
          fits = true;
          break;
        }
      }
    }
    if (!fits) {
      if (!d_upstream_done) return false;
      // end of a finite stream: run on what is left if the default contract (noutput new items) holds
      noutput = d_output_multiple > 1 ? d_output_multiple : 1;
      if (avail < noutput) {
        d_done.store(true, std::memory_order_release);
        return true;
      }
      if (d_output_multiple <= 1) {
        d_done.store(true, std::memory_order_release); // rate-changing blocks without an output multiple cannot use a partial tail
        return true;
      }
      d_draining = true;
    }
  }
  if ((int)d_out.size() < 2 * noutput + 16) d_out.resize(2 * noutput + 16);
  gr_vector_int ninput(1, (int)std::min<int64_t>(avail + (int64_t)history() - 1, INT32_MAX));
  gr_vector_const_void_star in(1, nullptr);
  gr_vector_void_star out(1, d_out.data());
  if (!source) in[0] = d_in.data() + (size_t)(d_nread - (int64_t)(history() - 1) - d_in_abs0);
  d_consumed = 0;
  d_new_tags.clear();
  const int64_t nread_before = d_nread;
  const int ret = general_work(noutput, ninput, in, out);

  // tag propagation for rate-1 blocks: same absolute offset (GNU Radio block_executor, rrate == 1)
  std::vector<tag_t> out_tags = d_new_tags;
  if (!source && d_tpp != TPP_DONT && d_consumed > 0)
    for (const tag_t &t : d_in_tags)
      if ((int64_t)t.offset >= nread_before && (int64_t)t.offset < nread_before + d_consumed) out_tags.push_back(t);

  if (!source && d_consumed > 0) {
    d_nread += d_consumed;
    d_backlog.fetch_sub((size_t)d_consumed, std::memory_order_relaxed);
    const int64_t keep_from = d_nread - (int64_t)(history() - 1);
    if (keep_from > d_in_abs0) {
      d_in.erase(d_in.begin(), d_in.begin() + (size_t)(keep_from - d_in_abs0));
      d_in_abs0 = keep_from;
    }
    d_in_tags.erase(std::remove_if(d_in_tags.begin(), d_in_tags.end(),
                                   [&](const tag_t &t) { return (int64_t)t.offset < d_nread; }),
                    d_in_tags.end());
  }
  if (ret > 0) {
    for (block *d : d_downstream)
      if (!d->d_done) d->accept(d_out.data(), (size_t)ret, out_tags);
    d_nwritten += (uint64_t)ret;
  }
  if (ret == WORK_DONE) {
    d_done.store(true, std::memory_order_release); // (after the hand-over: a downstream that sees it finds everything in its inbox)
    return true;
  }
  return ret > 0 || d_consumed > 0;
}

int sync_block::general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &input_items,
                             gr_vector_void_star &output_items)
{
  const int r = work(noutput_items, input_items, output_items);
  if (r > 0) consume_each(r);
  return r;
}

// ---------------------------------------------------------------- top_block
void top_block::add(block_sptr b)
{
  if (std::find(d_blocks.begin(), d_blocks.end(), b) == d_blocks.end()) d_blocks.push_back(b);
}

void top_block::connect(block_sptr src, block_sptr dst)
{
  if (dst->d_upstream) throw std::runtime_error("connect: input of " + dst->name() + " already connected");
  add(src);
  add(dst);
  src->d_downstream.push_back(dst.get());
  dst->d_upstream = src.get();
}

void top_block::msg_connect(basic_block_sptr src, const std::string &src_port, basic_block_sptr dst,
                            const std::string &dst_port)
{
  d_keepalive.push_back(src);
  d_keepalive.push_back(dst);
  src->d_out_ports[pmt::intern(src_port).get()].emplace_back(dst.get(), pmt::intern(dst_port));
}

void top_block::run(double max_seconds)
{
  const auto t0 = std::chrono::steady_clock::now();
  // order blocks so that every block comes after its upstream
  std::vector<block_sptr> order;
  std::vector<block_sptr> pending = d_blocks;
  while (!pending.empty()) {
    bool moved = false;
    for (auto it = pending.begin(); it != pending.end();) {
      block *up = (*it)->d_upstream;
      const bool ready = !up || std::any_of(order.begin(), order.end(), [&](const block_sptr &b) { return b.get() == up; });
      if (ready) {
        order.push_back(*it);
        it = pending.erase(it);
        moved = true;
      } else {
        ++it;
      }
    }
    if (!moved) throw std::runtime_error("top_block: cycle in flowgraph");
  }
  for (;;) {
    bool progress = false;
    for (auto &b : order) {
      if (b->d_upstream && b->d_upstream->d_done) b->d_upstream_done = true;
      while (b->step()) progress = true; // drain this block before moving downstream
    }
    if (!progress) break;
    if (max_seconds >= 0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= max_seconds)
      break;
  }
}

// ---------------------------------------------------------------- thread-per-block scheduler
void top_block::wake_all()
{
  {
    std::lock_guard<std::mutex> lock(d_wake_mu);
    ++d_wake_gen;
  }
  d_wake.notify_all();
}

void block::thread_main(top_block *tb)
{
  for (;;) {
    uint64_t gen;
    {
      std::lock_guard<std::mutex> lock(tb->d_wake_mu);
      gen = tb->d_wake_gen;
    }
    if (tb->d_stop.load(std::memory_order_acquire) || d_done.load(std::memory_order_acquire)) break;
    bool moved = false;
    while (!tb->d_stop.load(std::memory_order_acquire) && step()) moved = true;
    if (moved) {
      tb->wake_all();
      continue;
    }
    // nothing to do until a neighbour has moved: wait for the next signal (a bounded wait: a signal sent between the
    // generation read above and here is seen through the generation, one that the bookkeeping missed through the timeout)
    std::unique_lock<std::mutex> lock(tb->d_wake_mu);
    // (wait_until on the system clock = pthread_cond_timedwait; wait_for would be pthread_cond_clockwait, which GCC 11's
    // ThreadSanitizer does not intercept -- it then misses the unlock inside the wait and reports a double lock)
    tb->d_wake.wait_until(lock, std::chrono::system_clock::now() + std::chrono::milliseconds(2),
                          [&] { return tb->d_wake_gen != gen || tb->d_stop.load(std::memory_order_acquire); });
  }
  tb->wake_all();
}

void top_block::run_threaded(double max_seconds)
{
  const auto t0 = std::chrono::steady_clock::now();
  d_stop.store(false, std::memory_order_release);
  std::vector<std::thread> threads;
  std::vector<std::exception_ptr> errors(d_blocks.size());
  for (size_t i = 0; i < d_blocks.size(); ++i)
    threads.emplace_back([this, i, &errors] {
      try {
        d_blocks[i]->thread_main(this);
      } catch (...) {
        errors[i] = std::current_exception();
        d_stop.store(true, std::memory_order_release);
        wake_all();
      }
    });
  for (;;) {
    bool all_done = true;
    for (auto &b : d_blocks) all_done = all_done && b->done();
    if (all_done || d_stop.load(std::memory_order_acquire)) break;
    if (max_seconds >= 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= max_seconds) break;
    std::unique_lock<std::mutex> lock(d_wake_mu);
    d_wake.wait_until(lock, std::chrono::system_clock::now() + std::chrono::milliseconds(5));
  }
  d_stop.store(true, std::memory_order_release);
  wake_all();
  for (auto &t : threads) t.join();
  for (auto &e : errors)
    if (e) std::rethrow_exception(e);
}

// ---------------------------------------------------------------- stock blocks
namespace blocks {

vector_source_c::vector_source_c(const std::vector<gr_complex> &d, bool r)
    : sync_block("vector_source_c", io_signature::make(0, 0, 0), io_signature::make(1, 1, sizeof(gr_complex))),
      d_data(d), d_repeat(r)
{
}

vector_source_c::sptr vector_source_c::make(const std::vector<gr_complex> &data, bool repeat)
{
  return sptr(new vector_source_c(data, repeat));
}

int vector_source_c::work(int noutput_items, gr_vector_const_void_star &, gr_vector_void_star &output_items)
{
  gr_complex *out = static_cast<gr_complex *>(output_items[0]);
  if (d_data.empty()) return WORK_DONE;
  int n = 0;
  while (n < noutput_items) {
    if (d_pos >= d_data.size()) {
      if (!d_repeat) break;
      d_pos = 0;
    }
    const size_t k = std::min<size_t>((size_t)(noutput_items - n), d_data.size() - d_pos);
    std::memcpy(out + n, d_data.data() + d_pos, k * sizeof(gr_complex));
    n += (int)k;
    d_pos += k;
  }
  return n == 0 ? (int)WORK_DONE : n;
}

vector_source_c::sptr file_source(const std::string &filename, bool repeat)
{
  FILE *f = std::fopen(filename.c_str(), "rb");
  if (!f) throw std::runtime_error("file_source: cannot open " + filename);
  std::vector<gr_complex> data;
  gr_complex buf[4096];
  size_t n;
  while ((n = std::fread(buf, sizeof(gr_complex), 4096, f)) > 0) data.insert(data.end(), buf, buf + n);
  std::fclose(f);
  return vector_source_c::make(data, repeat);
}

head::head(uint64_t n)
    : sync_block("head", io_signature::make(1, 1, sizeof(gr_complex)), io_signature::make(1, 1, sizeof(gr_complex))),
      d_nitems(n)
{
}

int head::work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
{
  if (d_count >= d_nitems) return WORK_DONE;
  const uint64_t n = std::min<uint64_t>((uint64_t)noutput_items, d_nitems - d_count);
  std::memcpy(output_items[0], input_items[0], n * sizeof(gr_complex));
  d_count += n;
  return (int)n;
}

null_sink::null_sink()
    : sync_block("null_sink", io_signature::make(1, 1, sizeof(gr_complex)), io_signature::make(0, 0, 0))
{
}

message_debug::message_debug() : basic_block("message_debug")
{
  message_port_register_in(pmt::mp("store"));
  set_msg_handler(pmt::mp("store"), [this](pmt::pmt_t m) {
    std::lock_guard<std::mutex> l(d_mu);
    d_msgs.push_back(m);
  });
}

} // namespace blocks
} // namespace gr
