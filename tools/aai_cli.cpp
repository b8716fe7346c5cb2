// aai_cli.cpp -- command-line front end for CSV images: read, resample on the GPU through
// include/AreaAverageInterpolation.hpp, write <base>_mod.csv next to the input.
//
// It serves the users of the reference's driver (Source.cpp:1434-1599), which hard-codes its parameters and asks
// for the source to be edited (Source.cpp:1528-1534, README.md:19): the same parameters are flags here and default to
// the reference's values,
//   --input Test_film_dose.csv --src-res 150 --dst-res 25.4 --iso-x 455 --iso-y 455 --angle 1.5 --mode 2
// and what a user of that driver sees stays the same: the console messages, the ".csv" / ".CSV" check, the output
// name, and three properties of its CSV handling that files in the wild may rely on:
//   (1) a field that does not start with a number is skipped, not an error            (Source.cpp:1454-1468)
//   (2) an empty line yields an empty row                                              (Source.cpp:1482-1485)
//   (3) values are written with the stream's default 6 significant digits              (Source.cpp:1508)
// One deliberate difference: a row shorter than the widest row so far is padded with zeros (the reference reads
// past the end of its vector there, Source.cpp:1486-1488).
//
// build:  g++ -O2 -std=c++17 -Iinclude tools/aai_cli.cpp -o aai_cli -Larea_average_interpolation_amd -laai_hip \
//             -Wl,-rpath,'$ORIGIN/area_average_interpolation_amd'
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "AreaAverageInterpolation.hpp"

namespace {

struct Options {
    std::string input = "Test_film_dose.csv";     // Source.cpp:1529
    double srcRes = 150, dstRes = 25.4;            // Source.cpp:1530-1531
    double isoX = 455, isoY = 455;                 // Source.cpp:1532
    double angle = 1.5;                            // Source.cpp:1533
    int mode = 2;                                  // Source.cpp:1534: 1 area average, 2 fast area average
    bool exactAreas = false, quiet = false;
};

// directory (with its trailing separator), file stem and extension (with its dot).  A backslash anywhere makes the
// LAST backslash the separator, otherwise the last slash is; the extension starts at the last dot of the whole string.
struct PathParts {
    std::string dir, stem, ext;
    explicit PathParts(const std::string &full)
    {
        std::size_t sep = full.find_last_of('\\');
        if (sep == std::string::npos) sep = full.find_last_of('/');
        const std::size_t nameAt = sep == std::string::npos ? 0 : sep + 1;
        const std::size_t dot = full.find_last_of('.');
        dir = full.substr(0, nameAt);
        if (dot == std::string::npos) stem = full.substr(nameAt);
        else {
            ext = full.substr(dot);
            stem = dot >= nameAt ? full.substr(nameAt, dot - nameAt) : full.substr(nameAt);
        }
    }
    std::string sibling(const char *suffix) const { return dir + stem + suffix + ext; }
};

// the numbers of one text line; anything between two commas that does not begin with a number is left out
void parse_row(const std::string &text, std::vector<double> &values)
{
    values.clear();
    const char *p = text.c_str();
    const char *const end = p + text.size();
    for (;;) {
        const char *comma = static_cast<const char *>(std::memchr(p, ',', (std::size_t)(end - p)));
        const std::string field(p, comma ? comma : end);
        char *stop = nullptr;
        errno = 0;
        const double v = std::strtod(field.c_str(), &stop);
        if (stop != field.c_str() && errno != ERANGE) values.push_back(v);       // leading number, trailing text ignored
        if (!comma) break;
        p = comma + 1;
    }
}

bool read_csv(const std::string &file, IMG &image)
{
    std::ifstream in(file);
    if (!in) {
        std::cout << "Failed to read csv file." << std::endl;
        return false;
    }
    image.clear();
    std::vector<double> values;
    std::size_t widest = 0;
    for (std::string text; std::getline(in, text);) {
        parse_row(text, values);
        if (values.size() > widest) widest = values.size();
        if (!values.empty()) values.resize(widest, 0.0);      // pad short rows; an empty line stays an empty row
        image.push_back(values);
    }
    return true;
}

bool write_csv(const std::string &file, const IMG &image)
{
    std::ofstream out(file);
    if (!out) {
        std::cout << "Failed to write csv file." << std::endl;
        return false;
    }
    if (image.empty()) {
        std::cout << "There is no data in src array." << std::endl;
        std::cout << "Failed to write csv file." << std::endl;
        return false;
    }
    const std::size_t columns = image.front().size();       // every row is written with the first row's width
    for (const std::vector<double> &row : image) {
        const char *glue = "";
        for (std::size_t c = 0; c < columns; ++c) {
            out << glue << row[c];                           // default stream formatting: 6 significant digits
            glue = ",";
        }
        out << std::endl;
    }
    return true;
}

int abnormal()
{
    std::cout << "Run terminated abnormally." << std::endl;
    return -1;
}

// returns false (after printing why) when the command line is unusable
bool parse_command_line(int argc, char **argv, Options &o)
{
    struct Flag { const char *name; double *number; int *integer; std::string *text; bool *toggle; };
    const Flag flags[] = {
        {"--input", nullptr, nullptr, &o.input, nullptr},   {"--src-res", &o.srcRes, nullptr, nullptr, nullptr},
        {"--dst-res", &o.dstRes, nullptr, nullptr, nullptr}, {"--iso-x", &o.isoX, nullptr, nullptr, nullptr},
        {"--iso-y", &o.isoY, nullptr, nullptr, nullptr},     {"--angle", &o.angle, nullptr, nullptr, nullptr},
        {"--mode", nullptr, &o.mode, nullptr, nullptr},      {"--exact-areas", nullptr, nullptr, nullptr, &o.exactAreas},
        {"--quiet", nullptr, nullptr, nullptr, &o.quiet},
    };
    for (int i = 1; i < argc; ++i) {
        const Flag *hit = nullptr;
        for (const Flag &f : flags)
            if (std::strcmp(argv[i], f.name) == 0) hit = &f;
        if (!hit) {
            std::cout << "Unknown option " << argv[i] << std::endl;
            return false;
        }
        if (hit->toggle) { *hit->toggle = true; continue; }
        if (i + 1 >= argc) {
            std::cout << "Missing value for " << hit->name << std::endl;
            return false;
        }
        const char *value = argv[++i];
        if (hit->text) *hit->text = value;
        else if (hit->number) *hit->number = std::atof(value);
        else *hit->integer = std::atoi(value);
    }
    return true;
}

}  // namespace

int main(int argc, char **argv)
{
    Options opt;
    if (!parse_command_line(argc, argv, opt)) return -1;

    const PathParts where(opt.input);
    if (where.ext != ".csv" && where.ext != ".CSV") {
        std::cout << "As for the image format, only csv format can be used." << std::endl;
        std::cout << "* path  : " << where.dir << std::endl;
        std::cout << "* base  : " << where.stem << std::endl;
        std::cout << "* ext   : " << where.ext << std::endl;
        return abnormal();
    }
    IMG source, result;
    if (!read_csv(opt.input, source)) return abnormal();
    if (opt.mode != 1 && opt.mode != 2) {          // (the reference also reads the file first)
        std::cout << "Invalid interpolation mode is selected." << std::endl;
        std::cout << "Interpolation mode should be 1 or 2." << std::endl;
        std::cout << " * Selected interpolation mode : " << opt.mode << std::endl;
        return abnormal();
    }

    AreaAverageInterpolation engine;
    engine.verbose = !opt.quiet;
    engine.policy = opt.exactAreas ? AAI_POLICY_EXACT : AAI_POLICY_REFERENCE;
    const dP srcRes{opt.srcRes, opt.srcRes}, dstRes{opt.dstRes, opt.dstRes}, iso{opt.isoX, opt.isoY};
    dP resultIso;
    const auto t0 = std::chrono::steady_clock::now();
    const std::pair<bool, std::string> status =
        opt.mode == 1 ? engine.areaAverageInterpolation(source, result, srcRes, dstRes, iso, resultIso, opt.angle)
                      : engine.fastAreaAverageInterpolation(source, result, srcRes, dstRes, iso, resultIso, opt.angle);
    const std::chrono::duration<double, std::milli> spent = std::chrono::steady_clock::now() - t0;
    std::cout << "Calculation time : " << spent.count() << " [ms]" << std::endl;
    if (!status.first) {
        std::cout << status.second << std::endl;
        return abnormal();
    }
    if (!write_csv(where.sibling("_mod"), result)) return abnormal();
    std::cout << "Run terminated correctly." << std::endl;
    return 0;
}
