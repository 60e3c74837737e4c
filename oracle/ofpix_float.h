/* oracle/ofpix_float.h -- TEST INFRASTRUCTURE ONLY.
 * Selects the reference's float build without touching its sources: src/of.h:4-10 types ofpix_t as double when
 * OFPIX_DOUBLE is defined in that header and as float when the line is commented out.  Force-including this file
 * with -DOF_OF_H (the guard of src/of.h) gives the second variant (oracle variant 2). */
#ifndef OFX_OFPIX_FLOAT_H
#define OFX_OFPIX_FLOAT_H
typedef float ofpix_t;
#endif
