/* ccmp_split.h — what a split launch asks of the counting sort that precedes it: the cut of the descending order between the
 * latency blocks (front) and the throughput layout (rest), decided on the device from the sort's histogram and left in the
 * queue words the launches read — queue[4] the front's length, queue[0] the throughput kernel's first ticket, queue[3] its count
 * of finished units.  Decided by the sort's own kernel (round 5; a launch of its own, and one more to clear the words, until then:
 * two of the five kernel boundaries between the scout and the fork). */
#ifndef CCMP_SPLIT_H
#define CCMP_SPLIT_H
struct ccmp_split_req {
  unsigned long long *queue; /* nullptr: no split */
  int kind;                  /* 1: the units predicted >= p_low, at most `limit` (projector: fd_split_kernel's rule);
                                2: >= p_high where those carry permille / 1000 of the predicted work, else >= p_low (extend step: geo_split2_kernel's) */
  int p_low, p_high, permille;
  unsigned int limit;
  int clear;                 /* this many 64-bit words from queue[0] on are zeroed first (the others of the launch's block) */
};
#endif /* CCMP_SPLIT_H */
