// mt_driver.cpp -- thin C exports over the REFERENCE's own MT19937
// (lsb/cub/test/mersenne.h), compiled from where it lies under
// /root/reference into oracle/_ref/libref_mersenne.so.  It exists only to pin
// oracle.c's restatement of that generator and of RandomBits
// (lsb/cub/test/test_util.h:408-458) against the reference's code.
// TEST INFRASTRUCTURE ONLY.  No reference source is copied into this repo.
#include <mersenne.h>

extern "C" {
void ref_mt_init_by_array(unsigned int *key, int len) { mersenne::init_by_array(key, len); }
void ref_mt_init_genrand(unsigned int s) { mersenne::init_genrand(s); }
unsigned int ref_mt_genrand_int32(void) { return mersenne::genrand_int32(); }
}
