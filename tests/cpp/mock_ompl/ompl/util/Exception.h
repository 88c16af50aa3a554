// Interface mock (see ../../README.md)
#pragma once
#include <stdexcept>
#include <string>
namespace ompl {
class Exception : public std::runtime_error {
public:
  explicit Exception(const std::string &what) : std::runtime_error(what) {}
};
}  // namespace ompl
