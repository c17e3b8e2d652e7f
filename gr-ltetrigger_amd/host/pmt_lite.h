// Minimal polymorphic-type (PMT) stand-in: exactly the subset of GNU Radio's pmt API that the
// reference blocks use (intern / from_long / to_long / PMT_NIL / PMT_T / PMT_F / dict_add ...,
// lib/pss_impl.cc:39-40, lib/sss_impl.cc:38-43,141-150, lib/mib_impl.cc:185-251).  Compiled only
// when GNU Radio itself is absent; with GNU Radio present the block sources use <pmt/pmt.h>.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

namespace pmt {

struct pmt_base {
  enum kind_t { NIL, BOOL, LONG, SYMBOL, DICT } kind = NIL;
  bool b = false;
  long l = 0;
  std::string s;
  std::vector<std::pair<std::shared_ptr<const pmt_base>, std::shared_ptr<const pmt_base>>> d;
};
typedef std::shared_ptr<const pmt_base> pmt_t;

inline pmt_t make_(pmt_base::kind_t k)
{
  auto p = std::make_shared<pmt_base>();
  p->kind = k;
  return p;
}

inline const pmt_t &nil_()
{
  static const pmt_t v = make_(pmt_base::NIL);
  return v;
}
inline const pmt_t &true_()
{
  static const pmt_t v = [] { auto p = std::make_shared<pmt_base>(); p->kind = pmt_base::BOOL; p->b = true; return p; }();
  return v;
}
inline const pmt_t &false_()
{
  static const pmt_t v = [] { auto p = std::make_shared<pmt_base>(); p->kind = pmt_base::BOOL; p->b = false; return p; }();
  return v;
}
#define PMT_NIL ::pmt::nil_()
#define PMT_T ::pmt::true_()
#define PMT_F ::pmt::false_()

// interned symbols compare equal by pointer, like GNU Radio's
inline pmt_t intern(const std::string &s)
{
  static std::map<std::string, pmt_t> table;
  static std::mutex mu; // (blocks intern keys from their own threads: mib_impl::pack_cell, the cellstore's handler)
  std::lock_guard<std::mutex> lock(mu);
  auto it = table.find(s);
  if (it != table.end()) return it->second;
  auto p = std::make_shared<pmt_base>();
  p->kind = pmt_base::SYMBOL;
  p->s = s;
  table[s] = p;
  return p;
}
inline pmt_t mp(const std::string &s) { return intern(s); }
inline pmt_t from_long(long v)
{
  auto p = std::make_shared<pmt_base>();
  p->kind = pmt_base::LONG;
  p->l = v;
  return p;
}
inline long to_long(const pmt_t &p) { return p->l; }
inline bool is_null(const pmt_t &p) { return p->kind == pmt_base::NIL; }
inline bool is_symbol(const pmt_t &p) { return p->kind == pmt_base::SYMBOL; }
inline bool is_integer(const pmt_t &p) { return p->kind == pmt_base::LONG; }
inline bool is_dict(const pmt_t &p) { return p->kind == pmt_base::DICT; }
inline std::string symbol_to_string(const pmt_t &p) { return p->s; }
inline bool eq(const pmt_t &a, const pmt_t &b) { return a.get() == b.get(); }

inline pmt_t make_dict() { return make_(pmt_base::DICT); }
inline pmt_t dict_add(const pmt_t &dict, const pmt_t &key, const pmt_t &value)
{
  auto p = std::make_shared<pmt_base>(*dict);
  p->kind = pmt_base::DICT;
  for (auto &kv : p->d)
    if (kv.first.get() == key.get()) {
      kv.second = value;
      return p;
    }
  p->d.emplace_back(key, value);
  return p;
}
inline pmt_t dict_ref(const pmt_t &dict, const pmt_t &key, const pmt_t &not_found)
{
  for (auto &kv : dict->d)
    if (kv.first.get() == key.get()) return kv.second;
  return not_found;
}

// what pmt.to_python() + json.dumps would print for the subset above
inline std::string to_json(const pmt_t &p)
{
  std::ostringstream o;
  switch (p->kind) {
  case pmt_base::NIL: o << "null"; break;
  case pmt_base::BOOL: o << (p->b ? "true" : "false"); break;
  case pmt_base::LONG: o << p->l; break;
  case pmt_base::SYMBOL: o << '"' << p->s << '"'; break;
  case pmt_base::DICT: {
    o << '{';
    bool first = true;
    for (auto &kv : p->d) {
      if (!first) o << ", ";
      first = false;
      o << '"' << kv.first->s << "\": " << to_json(kv.second);
    }
    o << '}';
    break;
  }
  }
  return o.str();
}

} // namespace pmt
