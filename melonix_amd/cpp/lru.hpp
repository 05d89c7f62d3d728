// lru.hpp — a small least-recently-used table used by both facade classes.
// Semantics the melonix UI relies on: touching a key moves it to the front; inserting beyond
// `capacity` evicts from the back and hands the evicted (key, value) to the caller.
#pragma once
#include <cstddef>
#include <list>
#include <optional>
#include <unordered_map>
#include <utility>

namespace melonix {

template <class Key, class Value, class Hash = std::hash<Key>>
class LruTable {
  struct Node {
    Key key;
    Value value;
  };
  using Order = std::list<Node>;

public:
  explicit LruTable(std::size_t capacity) : capacity_(capacity) {}

  std::size_t size() const { return index_.size(); }
  bool full() const { return index_.size() >= capacity_; }

  // Value of `key` (made most recent), or nullptr.
  Value *touch(const Key &key) {
    auto it = index_.find(key);
    if (it == index_.end()) return nullptr;
    order_.splice(order_.begin(), order_, it->second);
    return &it->second->value;
  }
  // Value of `key` without changing its age, or nullptr.
  Value *peek(const Key &key) {
    auto it = index_.find(key);
    return it == index_.end() ? nullptr : &it->second->value;
  }
  // Inserts (key must be absent) as most recent; no eviction.
  Value &insert(const Key &key, Value value) {
    order_.push_front(Node{key, std::move(value)});
    index_[key] = order_.begin();
    return order_.front().value;
  }
  // Removes and returns the least recently used entry.
  std::optional<std::pair<Key, Value>> evictOldest() {
    if (order_.empty()) return std::nullopt;
    Node n = std::move(order_.back());
    index_.erase(n.key);
    order_.pop_back();
    return std::make_pair(std::move(n.key), std::move(n.value));
  }
  // Removes `key` if present.
  void erase(const Key &key) {
    auto it = index_.find(key);
    if (it == index_.end()) return;
    order_.erase(it->second);
    index_.erase(it);
  }
  void clear() {
    index_.clear();
    order_.clear();
  }

private:
  std::size_t capacity_;
  Order order_;
  std::unordered_map<Key, typename Order::iterator, Hash> index_;
};

}  // namespace melonix
