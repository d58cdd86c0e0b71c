// The host logic of rdf_jit.cpp that needs no device: a program's canonical signature -> the template argument it spells
// (the kernel compiled at run time is spec_kernel<that type> / gspec_kernel<that type>), and what is refused.
#include "mini_test.hpp"
#include "../../rust_dataframe_amd/csrc/rdf_jit.cpp"

using rdfk::gprog_type;
using rdfk::prog_type;

TEST(exact_program_signatures_spell_their_types) {
    std::string t;
    CHECK(prog_type("P:-;V:(3 (2 (4 (1 (3 c0d c1d) c2d) c3d) k0d) c0d);-;S:1", t));
    CHECK_EQ(t, std::string("rdfk::Prog<rdfk::None, rdfk::Bin<3, rdfk::Bin<2, rdfk::Bin<4, rdfk::Bin<1, rdfk::Bin<3, rdfk::Col<0, RDF_F64>, rdfk::Col<1, RDF_F64>>, "
                            "rdfk::Col<2, RDF_F64>>, rdfk::Col<3, RDF_F64>>, rdfk::Imm<0, RDF_F64>>, rdfk::Col<0, RDF_F64>>, rdfk::None, 1>"));
    CHECK(prog_type("P:(30 c0d k0d);V:(3 (1 {9 c1i} c0d) {9 c2l});-;S:0", t));
    CHECK_EQ(t, std::string("rdfk::Prog<rdfk::Bin<30, rdfk::Col<0, RDF_F64>, rdfk::Imm<0, RDF_F64>>, rdfk::Bin<3, rdfk::Bin<1, rdfk::Cast<9, rdfk::Col<1, RDF_I32>>, "
                            "rdfk::Col<0, RDF_F64>>, rdfk::Cast<9, rdfk::Col<2, RDF_I64>>>, rdfk::None, 0>"));
    CHECK(prog_type("P:-;V:(1 (3 [24 c0f] c1f) [26 [8 c2f]]);c3l;S:1", t));       // unary nodes, a second value
    CHECK_EQ(t, std::string("rdfk::Prog<rdfk::None, rdfk::Bin<1, rdfk::Bin<3, rdfk::Un<24, rdfk::Col<0, RDF_F32>>, rdfk::Col<1, RDF_F32>>, "
                            "rdfk::Un<26, rdfk::Un<8, rdfk::Col<2, RDF_F32>>>>, rdfk::Col<3, RDF_I64>, 1>"));
    // every element type's tag
    CHECK(prog_type("P:-;V:(1 (1 (1 (1 c0u c1j) c2a) (1 c3h c4s)) (1 c5t c6b));-;S:1", t));
    CHECK(t.find("RDF_U64") != std::string::npos && t.find("RDF_U32") != std::string::npos && t.find("RDF_I8") != std::string::npos && t.find("RDF_U8") != std::string::npos &&
          t.find("RDF_I16") != std::string::npos && t.find("RDF_U16") != std::string::npos && t.find("RDF_BOOL") != std::string::npos);
}

TEST(grouped_program_signatures) {
    std::string t;
    CHECK(gprog_type("G6;P:(35 c0i k0d);K:(1 (3 {2 c1a} k1i) {2 c2a});V:c3d;(3 c4d (2 k2d c5d));", t));
    CHECK_EQ(t, std::string("rdfk::GProg<6, rdfk::Bin<35, rdfk::Col<0, RDF_I32>, rdfk::Imm<0, RDF_F64>>, rdfk::Bin<1, rdfk::Bin<3, rdfk::Cast<2, rdfk::Col<1, RDF_I8>>, "
                            "rdfk::Imm<1, RDF_I32>>, rdfk::Cast<2, rdfk::Col<2, RDF_I8>>>, rdfk::Col<3, RDF_F64>, rdfk::Bin<3, rdfk::Col<4, RDF_F64>, rdfk::Bin<2, rdfk::Imm<2, RDF_F64>, rdfk::Col<5, RDF_F64>>>>"));
    CHECK(gprog_type("G2;P:-;K:c0l;V:c1d;", t));
    CHECK_EQ(t, std::string("rdfk::GProg<2, rdfk::None, rdfk::Col<0, RDF_I64>, rdfk::Col<1, RDF_F64>>"));
    CHECK(!gprog_type("G2;P:-;K:c0l;V:", t));                 // no value
    CHECK(!gprog_type("P:-;V:c0d;-;S:1", t));
}

TEST(what_is_not_an_exact_program_is_refused) {
    std::string t;
    CHECK(!prog_type("P:-;V:(A0 (A1 c0d k0d) c1d);-;S:1", t));   // a shape kernel's runtime-operator slots: the catalogs' business
    CHECK(!prog_type("P:-;V:(3 c0d c1d);-;S:", t));
    CHECK(!prog_type("P:-;V:(3 c0d c1d);-", t));
    CHECK(!prog_type("P:-;V:(3 c0d c1x);-;S:1", t));             // unknown type tag
    CHECK(!prog_type("P:-;V:(3 c0d);-;S:1", t));                 // a binary node with one operand
    CHECK(!prog_type("P:-;V:[24 c0d;-;S:1", t));
    CHECK(!prog_type("P:-;V:(3 c0d c1d);-;S:1 ", t));            // trailing text
    CHECK(!prog_type("", t));
    std::string deep = "P:-;V:";
    for (int i = 0; i < 40; ++i) deep += "[24 ";
    deep += "c0d";
    for (int i = 0; i < 40; ++i) deep += "]";
    CHECK(!prog_type((deep + ";-;S:1").c_str(), t));              // deeper than any program the compiler accepts
}

TEST(code_object_symbols) {
    // not an ELF image / truncated images are refused without reading past the buffer
    std::string name;
    CHECK(!rdfk::kernel_symbol(std::vector<char>(), "_ZN4rdfk11spec_kernel", name));
    CHECK(!rdfk::kernel_symbol(std::vector<char>(100, 'x'), "_ZN4rdfk11spec_kernel", name));
    std::vector<char> fake(sizeof(Elf64_Ehdr), 0);
    std::memcpy(fake.data(), ELFMAG, SELFMAG);
    ((Elf64_Ehdr*)fake.data())->e_shoff = 1 << 20;               // section headers beyond the end
    ((Elf64_Ehdr*)fake.data())->e_shnum = 4;
    CHECK(!rdfk::kernel_symbol(fake, "_ZN4rdfk11spec_kernel", name));
}

int main() { return run_all(); }
