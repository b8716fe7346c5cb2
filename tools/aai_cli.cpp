// aai_cli.cpp -- command-line driver mirroring the reference's main() (Source.cpp:1434-1599) on top of
// include/AreaAverageInterpolation.hpp: read a CSV image, resample it on the GPU, write <base>_mod.csv.
//
// The reference hard-codes its parameters and asks the user to edit the source (Source.cpp:1528-1534,
// README.md:19).  Here the same parameters are flags whose DEFAULTS are the reference's hard-coded values:
//   --input Test_film_dose.csv --src-res 150 --dst-res 25.4 --iso-x 455 --iso-y 455 --angle 1.5 --mode 2
// CSV behaviour follows the reference (Source.cpp:1449-1515): fields that do not parse as numbers are
// skipped, a blank line appends an empty row, output uses the default ostream precision (6 significant
// digits), the result goes to <path><base>_mod<ext>, only .csv/.CSV is accepted, and the same messages
// are printed.  One deviation: a row shorter than the widest row seen so far is padded with zeros (the
// reference reads past the end of its vector there, Source.cpp:1486-1488).
//
// build:  g++ -O2 -std=c++17 -Iinclude tools/aai_cli.cpp -o aai_cli -Larea_average_interpolation_amd -laai_hip \
//             -Wl,-rpath,'$ORIGIN/area_average_interpolation_amd'
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "AreaAverageInterpolation.hpp"

using namespace std;

static void splitPath(const string &fullPath, string &path, string &base, string &extension)
{
    size_t dotPos = fullPath.rfind(".");
    size_t delimiterPos = fullPath.rfind("\\");
    if (delimiterPos == string::npos) delimiterPos = fullPath.rfind("/");
    delimiterPos++;                                   // npos + 1 == 0: no directory part
    extension = (dotPos == string::npos) ? "" : fullPath.substr(dotPos);
    base = fullPath.substr(delimiterPos, dotPos - delimiterPos);
    path = fullPath.substr(0, delimiterPos);
}

static vector<double> splitFields(const string &line, char delimiter)
{
    vector<double> ret;
    size_t start = 0;
    while (true) {
        size_t pos = line.find(delimiter, start);
        string field = line.substr(start, pos == string::npos ? string::npos : pos - start);
        try { ret.emplace_back(stod(field)); }
        catch (const invalid_argument &) { /* skipped, Source.cpp:1457-1459 */ }
        catch (const out_of_range &) { /* the reference would terminate here; skip instead */ }
        if (pos == string::npos) break;
        start = pos + 1;
    }
    return ret;
}

static bool csvRead(const string &path, IMG &data)
{
    ifstream fin(path);
    if (!fin) { cout << "Failed to read csv file." << endl; return false; }
    string str;
    size_t width = 0;
    data.clear();
    while (getline(fin, str)) {
        data.resize(data.size() + 1);
        vector<double> vec = splitFields(str, ',');
        if (width < vec.size()) width = vec.size();
        if (vec.empty()) continue;
        for (size_t i = 0; i < width; ++i) data.back().emplace_back(i < vec.size() ? vec[i] : 0.0);
    }
    return true;
}

static bool csvWrite(const string &path, const IMG &data)
{
    ofstream fout(path);
    if (!fout) { cout << "Failed to write csv file." << endl; return false; }
    if (data.size() == 0) {
        cout << "There is no data in src array." << endl;
        cout << "Failed to write csv file." << endl;
        return false;
    }
    const size_t w = data.front().size();
    for (size_t i = 0; i < data.size(); ++i) {
        for (size_t j = 0; j < w; ++j) {
            fout << data[i][j];
            if (j + 1 < w) fout << ",";
        }
        fout << endl;
    }
    return true;
}

int main(int argc, char **argv)
{
    string inputPath = "Test_film_dose.csv";          // Source.cpp:1529
    double srcRes = 150, dstRes = 25.4;               // Source.cpp:1530-1531
    double isoX = 455, isoY = 455;                    // Source.cpp:1532
    double rotationAngle = 1.5;                       // Source.cpp:1533
    int interpolationMode = 2;                        // Source.cpp:1534 (1: area average, 2: fast area average)
    bool exactPolicy = false, verbose = true;
    for (int i = 1; i < argc; ++i) {
        auto need = [&](const char *flag) -> const char * {
            if (i + 1 >= argc) { cout << "Missing value for " << flag << endl; exit(-1); }
            return argv[++i];
        };
        if (!strcmp(argv[i], "--input")) inputPath = need("--input");
        else if (!strcmp(argv[i], "--src-res")) srcRes = atof(need("--src-res"));
        else if (!strcmp(argv[i], "--dst-res")) dstRes = atof(need("--dst-res"));
        else if (!strcmp(argv[i], "--iso-x")) isoX = atof(need("--iso-x"));
        else if (!strcmp(argv[i], "--iso-y")) isoY = atof(need("--iso-y"));
        else if (!strcmp(argv[i], "--angle")) rotationAngle = atof(need("--angle"));
        else if (!strcmp(argv[i], "--mode")) interpolationMode = atoi(need("--mode"));
        else if (!strcmp(argv[i], "--exact-areas")) exactPolicy = true;
        else if (!strcmp(argv[i], "--quiet")) verbose = false;
        else { cout << "Unknown option " << argv[i] << endl; return -1; }
    }

    string path, base, extension;
    splitPath(inputPath, path, base, extension);
    if (extension != ".csv" && extension != ".CSV") {
        cout << "As for the image format, only csv format can be used." << endl;
        cout << "* path  : " << path << endl;
        cout << "* base  : " << base << endl;
        cout << "* ext   : " << extension << endl;
        cout << "Run terminated abnormally." << endl;
        return -1;
    }

    IMG src, dst;
    if (!csvRead(inputPath, src)) { cout << "Run terminated abnormally." << endl; return -1; }

    AreaAverageInterpolation aa;
    aa.verbose = verbose;
    aa.policy = exactPolicy ? AAI_POLICY_EXACT : AAI_POLICY_REFERENCE;
    dP dstIsocenter;
    pair<bool, string> ret;
    auto start = chrono::system_clock::now();
    switch (interpolationMode) {
    case 1: ret = aa.areaAverageInterpolation(src, dst, {srcRes, srcRes}, {dstRes, dstRes}, {isoX, isoY}, dstIsocenter, rotationAngle); break;
    case 2: ret = aa.fastAreaAverageInterpolation(src, dst, {srcRes, srcRes}, {dstRes, dstRes}, {isoX, isoY}, dstIsocenter, rotationAngle); break;
    default:
        cout << "Invalid interpolation mode is selected." << endl;
        cout << "Interpolation mode should be 1 or 2." << endl;
        cout << " * Selected interpolation mode : " << interpolationMode << endl;
        cout << "Run terminated abnormally." << endl;
        return -1;
    }
    auto end = chrono::system_clock::now();
    double time = static_cast<double>(chrono::duration_cast<chrono::microseconds>(end - start).count() / 1000.0);
    cout << "Calculation time : " << time << " [ms]" << endl;

    if (!ret.first) {
        cout << ret.second << endl;
        cout << "Run terminated abnormally." << endl;
        return -1;
    }
    string outputPath = path + base + "_mod" + extension;
    if (!csvWrite(outputPath, dst)) { cout << "Run terminated abnormally." << endl; return -1; }
    cout << "Run terminated correctly." << endl;
    return 0;
}
