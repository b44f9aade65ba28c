/*
 * ringhip_ops.h -- opcodes of the element-wise kernel family.
 *
 * One opcode per function of the reference's ring/vec_ops.go (file:line of each in the comment).  Per-element
 * formula with x = p1[j], y = p2[j], z = p3[j] (output, read first by the "then" forms), s0/s1 scalars,
 * q = modulus; MRed/BRed/CRed/MForm/IMForm are the primitives of ring/modular_reduction.go.
 */
#ifndef RINGHIP_OPS_H
#define RINGHIP_OPS_H

enum rh_vec_opcode {
  RH_OP_ADD = 0,                          /* addvec                       vec_ops.go:7    z = CRed(x+y)                 */
  RH_OP_ADD_LAZY = 1,                     /* addlazyvec                   :31             z = x+y                       */
  RH_OP_SUB = 2,                          /* subvec                       :55             z = CRed(x+q-y)               */
  RH_OP_SUB_LAZY = 3,                     /* sublazyvec                   :79             z = x+q-y                     */
  RH_OP_NEG = 4,                          /* negvec                       :103            z = q-x                       */
  RH_OP_REDUCE = 5,                       /* reducevec                    :125            z = BRedAdd(x)                */
  RH_OP_REDUCE_LAZY = 6,                  /* reducelazyvec                :147            z = BRedAddLazy(x)            */
  RH_OP_MUL_LAZY = 7,                     /* mulcoeffslazyvec             :169            z = x*y (wrapping)            */
  RH_OP_MUL_LAZY_THEN_ADD_LAZY = 8,       /* mulcoeffslazythenaddlazyvec  :193            z += x*y (wrapping)           */
  RH_OP_MUL_BARRETT = 9,                  /* mulcoeffsbarrettvec          :217            z = BRed(x,y)                 */
  RH_OP_MUL_BARRETT_LAZY = 10,            /* mulcoeffsbarrettlazyvec      :241            z = BRedLazy(x,y)             */
  RH_OP_MUL_BARRETT_THEN_ADD = 11,        /* mulcoeffsthenaddvec          :265            z = CRed(z+BRed(x,y))         */
  RH_OP_MUL_BARRETT_THEN_ADD_LAZY = 12,   /* mulcoeffsbarrettthenaddlazyvec :289          z += BRed(x,y)                */
  RH_OP_MUL_MONT = 13,                    /* mulcoeffsmontgomeryvec       :313            z = MRed(x,y)                 */
  RH_OP_MUL_MONT_LAZY = 14,               /* mulcoeffsmontgomerylazyvec   :336            z = MRedLazy(x,y)             */
  RH_OP_MUL_MONT_THEN_ADD = 15,           /* mulcoeffsmontgomerythenaddvec :360           z = CRed(z+MRed(x,y))         */
  RH_OP_MUL_MONT_THEN_ADD_LAZY = 16,      /* mulcoeffsmontgomerythenaddlazyvec :383       z += MRed(x,y)                */
  RH_OP_MUL_MONT_LAZY_THEN_ADD_LAZY = 17, /* mulcoeffsmontgomerylazythenaddlazyvec :407   z += MRedLazy(x,y)            */
  RH_OP_MUL_MONT_THEN_SUB = 18,           /* mulcoeffsmontgomerythensubvec :431           z = CRed(z+(q-MRed(x,y)))     */
  RH_OP_MUL_MONT_THEN_SUB_LAZY = 19,      /* mulcoeffsmontgomerythensublazyvec :455       z += q-MRed(x,y)              */
  RH_OP_MUL_MONT_LAZY_THEN_SUB_LAZY = 20, /* mulcoeffsmontgomerylazythensublazyvec :479   z += 2q-MRedLazy(x,y)         */
  RH_OP_MUL_MONT_LAZY_THEN_NEG = 21,      /* mulcoeffsmontgomerylazythenNegvec :504       z = 2q-MRedLazy(x,y)          */
  RH_OP_ADD_LAZY_THEN_MUL_SCALAR_MONT = 22,        /* addlazythenmulscalarmontgomeryvec :529       z = MRed(x+y,s0)     */
  RH_OP_ADD_SCALAR_LAZY_THEN_MUL_SCALAR_MONT = 23, /* addscalarlazythenmulscalarmontgomeryvec :553 z = MRed(x+s0,s1)    */
  RH_OP_ADD_SCALAR = 24,                  /* addscalarvec                 :575            z = CRed(x+s0)                */
  RH_OP_ADD_SCALAR_LAZY = 25,             /* addscalarlazyvec             :597            z = x+s0                      */
  RH_OP_ADD_SCALAR_LAZY_THEN_NEG_TWO_MODULUS_LAZY = 26, /* addscalarlazythenNegTwoModuluslazyvec :619 z = s0+2q-x       */
  RH_OP_SUB_SCALAR = 27,                  /* subscalarvec                 :642            z = CRed(x+q-s0)              */
  RH_OP_MUL_SCALAR_MONT = 28,             /* mulscalarmontgomeryvec       :664            z = MRed(x,s0)                */
  RH_OP_MUL_SCALAR_MONT_LAZY = 29,        /* mulscalarmontgomerylazyvec   :686            z = MRedLazy(x,s0)            */
  RH_OP_MUL_SCALAR_MONT_THEN_ADD = 30,    /* mulscalarmontgomerythenaddvec :708           z = CRed(z+MRed(x,s0))        */
  RH_OP_MUL_SCALAR_MONT_THEN_ADD_SCALAR = 31, /* mulscalarmontgomerythenaddscalarvec :730 z = CRed(MRed(x,s1)+s0)       */
  RH_OP_SUB_THEN_MUL_SCALAR_MONT_TWO_MODULUS = 32, /* subthenmulscalarmontgomeryTwoModulusvec :752 z = MRed(2q-y+x,s0)  */
  RH_OP_MFORM = 33,                       /* mformvec                     :778            z = MForm(x)                  */
  RH_OP_MFORM_LAZY = 34,                  /* mformlazyvec                 :800            z = MFormLazy(x)              */
  RH_OP_IMFORM = 35,                      /* imformvec                    :822            z = IMForm(x)                 */
  RH_OP_ZERO = 36,                        /* ZeroVec                      :847            z = 0                         */
  RH_OP_MASK = 37,                        /* MaskVec                      :870            z = (x >> s0) & s1            */
  RH_OP_COUNT = 38
};

#endif
