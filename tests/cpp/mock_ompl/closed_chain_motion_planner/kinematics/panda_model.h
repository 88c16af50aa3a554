// Interface mock (see ../../README.md): the ArmModel fields adapter part 2 reads — name, index and the base frame t_wb
// (the reference's panda_model.h:7-23; TRAC-IK and RBDL members left out).
#pragma once
#include <Eigen/Dense>
#include <memory>
#include <string>
struct ArmModel {
  std::string name;
  int index;
  Eigen::Isometry3d t_7e;  // end effector frame
  Eigen::Isometry3d t_wb;  // base frame wrt world frame
  Eigen::Isometry3d t_o7;
  ArmModel()
  {
    t_7e.setIdentity();
    t_wb.setIdentity();
    t_o7.setIdentity();
  }
};
typedef std::shared_ptr<ArmModel> ArmModelPtr;
