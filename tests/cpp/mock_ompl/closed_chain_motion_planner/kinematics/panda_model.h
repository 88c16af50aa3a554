// Interface mock (see ../../README.md): the two ArmModel fields adapter part 2 reads.
#pragma once
#include <memory>
#include <string>
struct ArmModel {
  std::string name;
  int index;
};
typedef std::shared_ptr<ArmModel> ArmModelPtr;
