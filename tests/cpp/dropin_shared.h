// Shared by the two translation units of the drop-in link test: included AFTER the replacement headers in each.
#pragma once
int dropin_grow_tree(const ob::SpaceInformationPtr &si_, const vector<const ob::State *> &neighbors, const ob::State *new_state,
                     vector<vector<ob::State *>> *lists);
int dropin_grow_tree_batched(const ob::SpaceInformationPtr &si_, const vector<const ob::State *> &neighbors, const ob::State *new_state,
                             vector<vector<ob::State *>> *lists);
bool dropin_check_motion(const ob::SpaceInformationPtr &si_, const ob::State *s1, const ob::State *s2);
