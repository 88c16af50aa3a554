// Interface mock (see ../../README.md)
#pragma once
namespace ompl { namespace base {
class State {
public:
  virtual ~State() = default;
  template <class T> T *as() { return static_cast<T *>(this); }
  template <class T> const T *as() const { return static_cast<const T *>(this); }
};
} }
